#!/usr/bin/env python3
"""Micro-benchmark of individual C-ABI kernels at the C2 training-step shapes
(HIP events, interleaved rounds, median).  Usage: python tools/kbench.py [filter]"""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_eeg_fmri_amd import _hip, ops  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=20, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    meds = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        meds.append(a.elapsed_time(b) / iters * 1e3)
    return statistics.median(meds)


def linear_case(M, K, N, out_f32=False, residual=False, act="none"):
    x = torch.randn(M, K, device="cuda").to(BF)
    w = torch.randn(N, K, device="cuda") / math.sqrt(K)
    b = torch.randn(N, device="cuda")
    wf = torch.empty(N, 1, K, dtype=BF, device="cuda")
    _hip.call("mm_prep_conv_weight", w.view(N, K, 1).contiguous(), wf, None, N, K, 1, K, 0)
    res = torch.randn(M, N, device="cuda") if residual else None
    of = torch.empty(M, N, device="cuda") if out_f32 else None
    ob = None if out_f32 else torch.empty(M, N, dtype=BF, device="cuda")

    def fn():
        _hip.call("mm_conv1d_fwd", x, wf, 1, M, K, N, 1, 0, None, b, ops.ACT[act], res, None, 1, None, of, ob, None, 0.0, 0, None, None, 0)
    us = timeit(fn)
    fl = 2.0 * M * K * N
    print(f"linear M={M} K={K} N={N} f32={out_f32} res={residual} act={act}: {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s")


def conv1d_case(B, T, Cin, Cout, k):
    x = torch.randn(B, T, Cin, device="cuda").to(BF)
    w = torch.randn(Cout, Cin, k, device="cuda") / math.sqrt(Cin * k)
    wf = torch.empty(Cout, k, Cin, dtype=BF, device="cuda")
    _hip.call("mm_prep_conv_weight", w.contiguous(), wf, None, Cout, Cin, k, Cin, 0)
    of = torch.empty(B, T, Cout, device="cuda")
    stats = torch.zeros(32, 2, Cout, device="cuda")
    b = torch.randn(Cout, device="cuda")

    def fn():
        _hip.call("mm_conv1d_fwd", x, wf, B, T, Cin, Cout, k, k // 2, None, b, 0, None, None, 1, stats, of, None, None, 0.0, 0, None, None, 0)
    us = timeit(fn)
    fl = 2.0 * B * T * Cin * Cout * k
    print(f"conv1d B={B} T={T} Cin={Cin} Cout={Cout} k={k}: {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s")


def conv1d_wgrad_case(B, T, Cin, Cout, k):
    x = torch.randn(B, T, Cin, device="cuda").to(BF)
    dy = torch.randn(B, T, Cout, device="cuda").to(BF)
    ws = torch.zeros(8, Cout, k, Cin, device="cuda")
    db = torch.zeros(32, Cout, device="cuda")

    def fn():
        _hip.call("mm_conv1d_wgrad", dy, x, ws, db, B, T, Cin, Cout, k, k // 2, Cin, k * Cin, 1, Cin, 8, Cout * k * Cin, 0)
    us = timeit(fn)
    fl = 2.0 * B * T * Cin * Cout * k
    print(f"wgrad1d B={B} T={T} Cin={Cin} Cout={Cout} k={k}: {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s")


def conv3d_dims_case(B, D, H, W, Cin, Cout, dgrad=False):
    """layer-2-shaped forward at arbitrary volume dims (BASELINE config #4: 64x64x48 input -> 32x32x24 here)"""
    x = torch.randn(B, D, H, W, Cin, device="cuda").to(BF)
    w = torch.randn(Cout, Cin, 27, device="cuda") / math.sqrt(Cin * 27)
    wf = torch.empty(Cout, 27, Cin, dtype=BF, device="cuda")
    _hip.call("mm_prep_conv_weight", w.contiguous(), wf, None, Cout, Cin, 27, Cin, 0)
    wres = (Cin == 32 and Cout == 64) or dgrad           # the weight-resident kernel writes bf16 only; so do data gradients
    of = torch.empty(B, D, H, W, Cout, device="cuda", dtype=BF if wres else torch.float32)
    stats = None if dgrad else torch.zeros(32, 2, Cout, device="cuda")
    b = None if dgrad else torch.randn(Cout, device="cuda")

    def fn():
        _hip.call("mm_conv3d_fwd", x, wf, B, D, H, W, Cin, Cout, b, stats, None if wres else of, of if wres else None)
    us = graph_time(fn, n=10)
    fl = 2.0 * B * D * H * W * Cin * Cout * 27
    print(f"conv3d B={B} {D}x{H}x{W} Cin={Cin} Cout={Cout}: {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s "
          f"({fl / us / 1e6 / 2500:.3f} of 2.5 PF; graph-replayed)")


def wres_sustained_case(B, D, H, W, launches=400):
    """the roofline kernel (layer 2, 32 -> 64 channels) as `launches` back-to-back eager launches with no host
    synchronisation in between - sustained clocks, the condition bench.py's roofline_c2_standalone / roofline_c4 time it
    under; run under `rocprofv3 --kernel-trace --stats` for the per-launch mean the bench line must agree with"""
    Cin, Cout = 32, 64
    x = torch.randn(B, D, H, W, Cin, device="cuda").to(BF)
    w = torch.randn(Cout, Cin, 27, device="cuda") / math.sqrt(Cin * 27)
    wf = torch.empty(Cout, 27, Cin, dtype=BF, device="cuda")
    _hip.call("mm_prep_conv_weight", w.contiguous(), wf, None, Cout, Cin, 27, Cin, 0)
    of = torch.empty(B, D, H, W, Cout, device="cuda", dtype=BF)
    stats = torch.zeros(32, 2, Cout, device="cuda")
    b = torch.randn(Cout, device="cuda")
    for _ in range(10):
        _hip.call("mm_conv3d_fwd", x, wf, B, D, H, W, Cin, Cout, b, stats, None, of)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(4.0e6))
    a.record()
    for _ in range(launches):
        _hip.call("mm_conv3d_fwd", x, wf, B, D, H, W, Cin, Cout, b, stats, None, of)
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / launches * 1e3
    fl = 2.0 * B * D * H * W * Cin * Cout * 27
    print(f"conv3d_wres sustained B={B} {D}x{H}x{W}: {launches} launches, {us:8.2f} us each  {fl / us / 1e6:8.1f} TF/s "
          f"({fl / us / 1e6 / 2500:.3f} of 2.5 PF; HIP events, gaps included)")


def conv3d_case(B, S, Cin, Cout, wgrad=True, fwd=True):
    x = torch.randn(B, S, S, S, Cin, device="cuda").to(BF)
    w = torch.randn(Cout, Cin, 27, device="cuda") / math.sqrt(Cin * 27)
    wf = torch.empty(Cout, 27, Cin, dtype=BF, device="cuda")
    _hip.call("mm_prep_conv_weight", w.contiguous(), wf, None, Cout, Cin, 27, Cin, 0)
    wres = Cin == 32 and Cout == 64                      # the weight-resident kernel writes bf16 only
    of = torch.empty(B, S, S, S, Cout, device="cuda", dtype=BF if wres else torch.float32)
    stats = torch.zeros(32, 2, Cout, device="cuda")
    b = torch.randn(Cout, device="cuda")

    def fn():
        _hip.call("mm_conv3d_fwd", x, wf, B, S, S, S, Cin, Cout, b, stats, None if wres else of, of if wres else None)
    fl = 2.0 * B * S ** 3 * Cin * Cout * 27
    if fwd:
        us = graph_time(fn)
        print(f"conv3d B={B} {S}^3 Cin={Cin} Cout={Cout}: {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s (graph-replayed)")
    if not wgrad:
        return
    dy = torch.randn(B, S, S, S, Cout, device="cuda").to(BF)
    import ctypes
    n = ctypes.c_int(0)
    _hip.call("mm_conv3d_wgrad_slots", B, S, S, S, Cin, Cout, ctypes.addressof(n))
    ws = torch.zeros(n.value, Cout, 27, Cin, device="cuda")           # slot mode, as in the training step

    def fn2():
        _hip.call("mm_conv3d_wgrad", dy, x, ws, None, B, S, S, S, Cin, Cout, Cin, 27 * Cin, 1, Cin, n.value, Cout * 27 * Cin, 1)
    us = graph_time(fn2)
    if os.environ.get("W3_DBG"):
        torch.cuda.synchronize()
        v = ws.view(n.value, -1)[:, :4].cpu()
        print(f"    cycles (kd = 2 workgroups): set-up + ring prologue {v[:, 0].mean():.0f} | tile loop done {v[:, 1].mean():.0f} | "
              f"K-split sum done {v[:, 2].mean():.0f} | stores drained {v[:, 3].mean():.0f}")
    print(f"wgrad3d B={B} {S}^3 Cin={Cin} Cout={Cout}: {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s ({fl / us / 1e6 / 2500:.3f} of 2.5 PF; "
          f"{n.value} slots, graph-replayed)")


def stamp_case(B, D, H, W):
    """in-kernel cycle stamps of a WRES_STAMPS build (tools/abl_build.sh s0; MMEEG_HIP_LIB=.../abl_s0.so)"""
    Cin, Cout = 32, 64
    x = torch.randn(B, D, H, W, Cin, device="cuda").to(BF)
    w = torch.randn(Cout, Cin, 27, device="cuda") / math.sqrt(Cin * 27)
    wf = torch.empty(Cout, 27, Cin, dtype=BF, device="cuda")
    _hip.call("mm_prep_conv_weight", w.contiguous(), wf, None, Cout, Cin, 27, Cin, 0)
    of = torch.empty(B, D, H, W, Cout, device="cuda", dtype=BF)
    stats = torch.zeros(32 * 2 * Cout + 256 * 16, device="cuda")
    b = torch.randn(Cout, device="cuda")
    for _ in range(20):
        _hip.call("mm_conv3d_fwd", x, wf, B, D, H, W, Cin, Cout, b, stats, None, of)
    torch.cuda.synchronize()
    st = stats[32 * 2 * Cout:].view(256, 16).cpu().double()
    k, tot, ticks, tiles = st[:, 0], st[:, 1], st[:, 2], st[:, 3]
    ghz = (tot / (ticks / 100.0)).mean() / 1e3
    print(f"stamps B={B} {D}x{H}x{W}: tiles/WG {tiles.mean():.1f}; kernel {tot.mean():.0f} cyc = {(ticks / 100).mean():.2f} us "
          f"({ghz:.2f} GHz); K loops {k.mean():.0f} cyc = {(k / tiles).mean():.0f} per tile (6912 = MFMA-bound); "
          f"outside K loops {(tot - k).mean():.0f} cyc = {((tot - k) / tiles).mean():.0f} per tile; per-WG kernel cyc min {tot.min():.0f} max {tot.max():.0f}")
    b0, e0 = st[:, 12], st[:, 13]
    if (e0.max() > b0.min()):
        print(f"    absolute 100 MHz clock: first start -> last end {(e0.max() - b0.min()) / 100:.2f} us; start stagger {(b0.max() - b0.min()) / 100:.2f} us; "
              f"end stagger {(e0.max() - e0.min()) / 100:.2f} us")
    names = ["halo 0 prefetch issued", "weight DMA issued", "halo constants parked", "first boundary done (halo 0 + W in LDS)",
             "K loop 1 issued", "K loop 2 issued", "last K loop issued", "flush issued"]
    for i, nm in enumerate(names):
        c = st[:, 4 + i]
        print(f"    {nm:40s} {c.mean():8.0f} cyc (min {c.min():.0f} max {c.max():.0f})")


def stream_stamp_case(B, D, H, W, Cin, Cout):
    """in-kernel cycle stamps of a STREAM_STAMPS build (tools/abl_stream.sh s0; MMEEG_HIP_LIB=.../sabl_0-DSTREAM_STAMPS.so)"""
    x = torch.randn(B, D, H, W, Cin, device="cuda").to(BF)
    w = torch.randn(Cout, Cin, 27, device="cuda") / math.sqrt(Cin * 27)
    wf = torch.empty(Cout, 27, Cin, dtype=BF, device="cuda")
    _hip.call("mm_prep_conv_weight", w.contiguous(), wf, None, Cout, Cin, 27, Cin, 0)
    of = torch.empty(B, D, H, W, Cout, device="cuda", dtype=BF)
    ntiles = B * ((D + 1) // 2) * ((H + 7) // 8) * ((W + 7) // 8)
    stats = torch.zeros(32 * 2 * Cout + ntiles * 8, device="cuda")
    for _ in range(20):
        _hip.call("mm_conv3d_fwd", x, wf, B, D, H, W, Cin, Cout, None, stats, None, of)
    torch.cuda.synchronize()
    st = stats[32 * 2 * Cout:].view(ntiles, 8).cpu().double()
    names = ["halo loads + writes issued", "halo + first stages in LDS (barrier)", "K loop done", "K-groups summed", "outputs stored", "stores drained"]
    print(f"stream stamps B={B} {D}x{H}x{W} {Cin}->{Cout}: {ntiles} workgroups; first start -> last end {(st[:, 7].max() - st[:, 6].min()) / 100:.2f} us; "
          f"start stagger {(st[:, 6].max() - st[:, 6].min()) / 100:.2f} us; mean workgroup {((st[:, 7] - st[:, 6]) / 100).mean():.2f} us")
    for i, nm in enumerate(names):
        c = st[:, i]
        print(f"    {nm:40s} {c.mean():8.0f} cyc (min {c.min():.0f} max {c.max():.0f})")


def attn_case(B=32, L=512, H=4, p=0.1):
    E = H * 32
    qkv = (torch.randn(B, L, 3 * E, device="cuda") * 0.5).to(BF)
    out = torch.empty(B, L, E, dtype=BF, device="cuda")
    lse = torch.empty(B, H, L, device="cuda")
    dout = torch.randn(B, L, E, device="cuda").to(BF)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, L, device="cuda")
    sc = 1 / math.sqrt(32)
    f = timeit(lambda: _hip.call("mm_attn_fwd", qkv, out, lse, B, L, H, 32, sc, p, 77, None, None, 0))
    b = timeit(lambda: _hip.call("mm_attn_bwd", qkv, out, dout, lse, dqkv, delta, B, L, H, 32, sc, p, 77, None, None, 0))
    fl = 4.0 * B * H * L * L * 32
    print(f"attention B={B} L={L} H={H} p={p}: fwd {f:6.1f} us ({fl / f / 1e6:6.1f} TF/s)   bwd (dq + dkv) {b:6.1f} us")


def floor_case():
    x = torch.zeros(64, device="cuda")
    y = torch.zeros(64, dtype=BF, device="cuda")
    us = timeit(lambda: _hip.call("mm_cast_bf16", x, y, 64))
    print(f"harness floor (64-element cast through _hip.call): {us:8.1f} us")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _hip.call("mm_cast_bf16", x, y, 64)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(20):
            _hip.call("mm_cast_bf16", x, y, 64)
    us = timeit(g.replay, iters=5) / 20
    print(f"same, 20 launches per hipGraph replay: {us:8.1f} us per kernel")


def graph_time(fn, n=20):
    """per-launch time of fn when n launches are replayed from one hipGraph"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return timeit(g.replay, iters=5) / n


def bn_bwd_case(R, S, N, pool, f32_dout):
    """the two BatchNorm+act backward passes at an EEG conv layer's shape (graph-replayed launches)"""
    y = torch.randn(R, S, N, device="cuda")
    out4 = torch.stack([torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda") * 0.1,
                        torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")]).contiguous()
    So = S // pool
    dout = torch.randn(R, So, N, device="cuda")
    db = None if f32_dout else dout.to(BF)
    df = dout if f32_dout else None
    sums = torch.zeros(32, 2, N, device="cuda")
    dy = torch.empty(R, S, N, dtype=BF, device="cuda")
    args = (R, S, N, ops.ACT["gelu"], pool, 0, 0.3, 1234, 0.0, 0, None)
    red = graph_time(lambda: _hip.call("mm_bn_act_bwd_reduce", y, out4, db, df, sums, *args))
    app = graph_time(lambda: _hip.call("mm_bn_act_bwd_apply", y, out4, db, df, sums, dy, None, *args, 1, 32))
    mb = (y.numel() * 4 + dout.numel() * (4 if f32_dout else 2)) / 1e6
    print(f"bn_bwd R={R} S={S} N={N} pool={pool} dout={'f32' if f32_dout else 'bf16'}: reduce {red:6.1f} us "
          f"({mb / red:5.2f} TB/s of {mb:.1f} MB)  apply {app:6.1f} us")


def split_case():
    """does running two half-batch chains on two streams beat one full-batch chain?  (linear + attention)"""
    import math as _m

    def mk(M):
        B = M // 512
        x = torch.randn(M, 128, device="cuda").to(BF)
        w = torch.randn(384, 128, device="cuda") / _m.sqrt(128)
        wf = torch.empty(384, 1, 128, dtype=BF, device="cuda")
        _hip.call("mm_prep_conv_weight", w.view(384, 128, 1).contiguous(), wf, None, 384, 128, 1, 128, 0)
        qkv = torch.empty(M, 384, dtype=BF, device="cuda")
        o = torch.empty(M, 128, dtype=BF, device="cuda")
        lse = torch.empty(B, 4, 512, device="cuda")
        w2 = torch.randn(128, 128, device="cuda") / _m.sqrt(128)
        wf2 = torch.empty(128, 1, 128, dtype=BF, device="cuda")
        _hip.call("mm_prep_conv_weight", w2.view(128, 128, 1).contiguous(), wf2, None, 128, 128, 1, 128, 0)
        res = torch.randn(M, 128, device="cuda")
        out = torch.empty(M, 128, device="cuda")

        def chain():
            for _ in range(4):
                _hip.call("mm_conv1d_fwd", x, wf, 1, M, 128, 384, 1, 0, None, None, 0, None, None, 1, None, None, qkv, None, 0.0, 0, None, None, 0)
                _hip.call("mm_attn_fwd", qkv, o, lse, B, 512, 4, 32, 1.0 / _m.sqrt(32), 0.1, 5, None, None, 0)
                _hip.call("mm_conv1d_fwd", o, wf2, 1, M, 128, 128, 1, 0, None, None, 0, res, None, 1, None, out, None, None, 0.1, 7, None, None, 0)
        return chain
    full = mk(16384)
    h1, h2 = mk(8192), mk(8192)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def two():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            h1()
        with torch.cuda.stream(s2):
            h2()
        cur.wait_stream(s1); cur.wait_stream(s2)
    for name, fn in (("one stream, B=32", full), ("one stream, B=16 twice", lambda: (h1(), h2())), ("two streams, B=16 each", two)):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        print(f"{name:28s} {timeit(g.replay, iters=10):8.1f} us  (4 x [QKV linear, attention fwd, out-proj])")


def l1_case(B=32, D=32, H=32, W=32, p=0.3):
    """the three training passes of the fused first voxel layer (conv3d_l1.hip) at the C2 shape: statistics (mode 0),
    forward (mode 1), one-pass backward (mode 4 + combine).  FLOPs per pass = 2 * 27 * 32 * voxels = 1.81 GF at C2;
    compulsory bytes: 4.2 MB fp32 volume in (+ 8.4 MB bf16 pooled output for the forward / pooled gradient in for the backward)."""
    import multimodal_eeg_fmri_amd.fmri_utils as Fm
    torch.manual_seed(0)
    enc = Fm.fMRIVolumeEncoder3D(1, 64, dropout=p).cuda().train()
    conv, bn = enc.conv_layers[0], enc.conv_layers[1]
    x = torch.randn(B, 1, D, H, W, device="cuda")
    out, s = ops.conv3d_l1_bn_act(x, conv, bn, training=True, drop_p=p)
    wimg, out4 = s["wimg"], s["out4"]
    stats = torch.zeros(32, 2, 32, device="cuda")
    dout = torch.randn(out.shape, device="cuda").to(BF)
    sums, a1 = torch.zeros(32, 2, 32, device="cuda"), torch.zeros(32, 27, 32, device="cuda")
    gram = torch.zeros(32, 32, 32, device="cuda")
    gramc = s["gramc"].clone()
    tapsum = torch.zeros(32, 32, device="cuda")
    dw, db = torch.zeros(32, 1, 3, 3, 3, device="cuda"), torch.zeros(32, device="cuda")
    fl = 2.0 * 27 * 32 * B * D * H * W
    tg = timeit(lambda: _hip.call("mm_conv3d_l1_gram", x, wimg, conv.bias, gram, stats, B, D, H, W))
    t0 = timeit(lambda: _hip.call("mm_conv3d_l1", 0, x, wimg, conv.bias, None, None, None, stats, None, None, None,
                                  B, D, H, W, 1, 0.0, 0, None))
    t1 = timeit(lambda: _hip.call("mm_conv3d_l1", 1, x, wimg, conv.bias, out4, None, None, None, out, None, None,
                                  B, D, H, W, 1, float(p), 123, None))
    t4 = timeit(lambda: _hip.call("mm_conv3d_l1_bwd", x, wimg, conv.bias, out4, dout, sums, a1, gramc, dw, db,
                                  B, D, H, W, 1, float(p), 123, None))
    tt = timeit(lambda: _hip.call("mm_conv3d_l1_tapsum", x, tapsum, B, D, H, W))
    inb, outb = x.numel() * 4, out.numel() * 2
    for name, t, byts in (("Gram matrix + BatchNorm sums", tg, inb),
                          ("stats by recompute (mode 0, ABI)", t0, inb), ("forward (mode 1)", t1, inb + outb),
                          ("backward (mode 4 + combine)", t4, inb + outb), ("tap sums (ABI)", tt, inb)):
        print(f"conv3d_l1 {name:34s} B={B} {D}x{H}x{W}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s (of 157 fp32 / 2500 bf16)  "
              f"{byts / 1e6:5.1f} MB compulsory -> {byts / t / 1e6:6.2f} TB/s of 8")


def step_case(config="c2", steps=4):
    """the captured training step itself, a few replays: the PMC collector (profiles/run_pmc_kernels.sh <tag> step "")
    reads counters for EVERY kernel of the step (the profiler serialises them: stand-alone figures in the step's order)"""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    torch.manual_seed(0)
    enc = None
    if config == "c5":
        from multimodal_eeg_fmri_amd.crossmodal_v4_enhancements import MultiScaleSTFTPowerEncoder
        enc = MultiScaleSTFTPowerEncoder(64, (64, 128), 32, 128, 2, 4, 0.3)
    tr = BridgeTrainer(eeg_channels=64, dropout=0.3, eeg_encoder=enc).train()
    eeg, fmri = synthetic_pairs(32, 64, 1024, (32, 32, 32), seed=1234)
    for _ in range(steps):
        out = tr.train_step(eeg, fmri)
    torch.cuda.synchronize()
    print(f"step case {config}: {steps} steps, loss {out['loss'].item():.4f}")


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    if flt in ("step", "step5"):
        step_case("c5" if flt == "step5" else "c2")
        return
    if flt in ("l1", "pmcl1"):
        l1_case()
        return
    if flt == "sus4":
        wres_sustained_case(32, 32, 32, 24)
        return
    if flt == "sus2":
        wres_sustained_case(32, 16, 16, 16, launches=1200)
        return
    if "floor" in flt:
        floor_case()
    M = 32 * 512
    if "lin" in flt or not flt:
        linear_case(M, 128, 384)
        linear_case(M, 128, 128, out_f32=True, residual=True)
        linear_case(M, 128, 512, act="gelu")
        linear_case(M, 512, 128, out_f32=True, residual=True)
        linear_case(M, 384, 128)
    if "conv1" in flt or not flt:
        conv1d_case(32, 1024, 64, 64, 7)
        conv1d_case(32, 1024, 64, 128, 5)
        conv1d_case(32, 512, 128, 128, 3)
        conv1d_wgrad_case(32, 1024, 64, 64, 7)
        conv1d_wgrad_case(32, 1024, 64, 128, 5)
        conv1d_wgrad_case(32, 512, 128, 128, 3)
        conv1d_wgrad_case(1, M, 128, 512, 1)
    if "split" in flt:
        split_case()
        return
    if "bn" in flt:
        bn_bwd_case(32, 512, 128, 1, True)
        bn_bwd_case(32, 1024, 128, 2, False)
        bn_bwd_case(32, 1024, 64, 1, False)
        return
    if "pmc3d" in flt:
        conv3d_case(32, 16, 32, 64, wgrad=False)
        return
    if flt == "stamp":
        stamp_case(32, 32, 32, 24)
        stamp_case(32, 16, 16, 16)
        return
    if "c4b" in flt:                # the two roofline shapes only (ablation sweeps)
        conv3d_dims_case(32, 32, 32, 24, 32, 64)
        conv3d_dims_case(32, 16, 16, 16, 32, 64)
        return
    if "pmc4" in flt:               # BASELINE config #4: layer 2 at 32 x 32 x 24
        conv3d_dims_case(32, 32, 32, 24, 32, 64)
        return
    if "attn" in flt:
        for p_ in (0.0, 0.1, 0.3):
            attn_case(p=p_)
        return
    if "c4" in flt:                 # full-resolution fMRI (64x64x48): layer 2 runs at 32x32x24
        for B in (4, 8, 32):
            conv3d_dims_case(B, 32, 32, 24, 32, 64)
        conv3d_dims_case(32, 16, 16, 16, 32, 64)
        return
    if flt == "pmcs":               # layer 3 forward only (PMC passes of the streaming kernel)
        conv3d_dims_case(32, 8, 8, 8, 64, 128)
        return
    if flt == "sstamp":
        stream_stamp_case(32, 8, 8, 8, 64, 128)
        stream_stamp_case(32, 8, 8, 8, 128, 64)
        stream_stamp_case(32, 16, 16, 16, 64, 32)
    if "stream" in flt:
        # conv3d_stream.hip: layer 3 forward, its data gradient, layer 2's data gradient (C2, then config #4 volumes)
        conv3d_dims_case(32, 8, 8, 8, 64, 128)
        conv3d_dims_case(32, 8, 8, 8, 128, 64, dgrad=True)
        conv3d_dims_case(32, 16, 16, 16, 64, 32, dgrad=True)
        conv3d_dims_case(32, 16, 16, 12, 64, 128)
        conv3d_dims_case(32, 16, 16, 12, 128, 64, dgrad=True)
        conv3d_dims_case(32, 32, 32, 24, 64, 32, dgrad=True)
    if flt == "wgrad3":
        conv3d_case(32, 16, 32, 64, fwd=False)
        conv3d_case(32, 8, 64, 128, fwd=False)
        return
    if "conv3" in flt or not flt:
        conv3d_case(32, 16, 32, 64)
        conv3d_case(32, 8, 64, 128)


if __name__ == "__main__":
    main()

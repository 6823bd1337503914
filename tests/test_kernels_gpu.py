"""GPU: kernel-level parity of the C-ABI entry points against plain fp32 torch
on the CPU (operands pre-rounded to bf16 where the kernel's MFMA operands are
bf16, so the only remaining difference is fp32 accumulation order)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import ref_functional as RF

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


def _hip():
    from multimodal_eeg_fmri_amd import _hip
    _hip.load()
    return _hip


def _stat(ws):
    """activation-statistics accumulator workspace (32, ...) -> fp32 sums"""
    from multimodal_eeg_fmri_amd.ops import ACC_STAT, acc_decode
    return acc_decode(ws, ACC_STAT).float()


def _grad(ws):
    """gradient accumulator workspace (32, ...) -> fp32 sums"""
    from multimodal_eeg_fmri_amd.ops import ACC_GRAD, acc_decode
    return acc_decode(ws, ACC_GRAD).float()


def _prep_w(hip, w, cinp, coutp=None):
    cout, cin, k = w.shape
    wf = torch.empty(cout, k, cinp, dtype=torch.bfloat16, device="cuda")
    wd = None
    if coutp:
        wd = torch.empty(cinp, k, coutp, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_prep_conv_weight", w.cuda().contiguous(), wf, wd, cout, cin, k, cinp, coutp or 0)
    return wf, wd


def _cpad(C):
    cp = max(16, (C + 15) // 16 * 16)
    return cp if cp <= 32 else (cp + 31) // 32 * 32


@pytest.mark.parametrize("B,C,T,Cout,k", [(2, 64, 256, 64, 7), (3, 64, 200, 128, 5), (2, 128, 128, 128, 3),
                                          (8, 8, 256, 64, 7), (1, 128, 96, 384, 1), (2, 512, 64, 128, 1),
                                          (1, 16, 40, 48, 7)])
def test_conv1d_fwd_matches_torch(B, C, T, Cout, k):
    hip = _hip()
    g = torch.Generator().manual_seed(B * 1000 + C + T + Cout + k)
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(Cout, C, k, generator=g) / math.sqrt(C * k)
    bias = torch.randn(Cout, generator=g)
    cp = _cpad(C)
    xg = torch.empty(B, T, cp, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pack_nct_bf16", x.cuda(), xg, B, C, T, cp)
    want_pack = torch.zeros(B, T, cp)
    want_pack[:, :, :C] = _bf(x).transpose(1, 2)
    assert torch.equal(xg.float().cpu(), want_pack)
    wf, _ = _prep_w(hip, w, cp)
    out = torch.empty(B, T, Cout, dtype=torch.float32, device="cuda")
    hip.call("mm_conv1d_fwd", xg, wf, B, T, cp, Cout, k, k // 2, None, bias.cuda(), 0, None, None, 1,
             None, out, None, None, 0.0, 0, None, None, 0)
    want = F.conv1d(_bf(x), _bf(w), bias, padding=k // 2).transpose(1, 2)
    torch.testing.assert_close(out.cpu(), want, rtol=1e-4, atol=1e-4)


def test_conv1d_fwd_epilogue_bn_gelu_pool_stats():
    hip = _hip()
    g = torch.Generator().manual_seed(5)
    B, C, T, Cout, k = 2, 64, 128, 128, 5
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(Cout, C, k, generator=g) / math.sqrt(C * k)
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = torch.randn(Cout, generator=g) * 0.3
    xg = torch.empty(B, T, C, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pack_nct_bf16", x.cuda(), xg, B, C, T, C)
    wf, _ = _prep_w(hip, w, C)
    out = torch.empty(B, T // 2, Cout, dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(32, 2, Cout, device="cuda")
    hip.call("mm_conv1d_fwd", xg, wf, B, T, C, Cout, k, k // 2, scale.cuda(), shift.cuda(), 1, None, None, 2,
             stats, None, out, None, 0.0, 0, None, None, 0)
    stats = _stat(stats)
    z = F.conv1d(_bf(x), _bf(w), None, padding=k // 2) * scale[None, :, None] + shift[None, :, None]
    want = F.max_pool1d(F.gelu(z), 2).transpose(1, 2)
    torch.testing.assert_close(out.float().cpu(), want, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(stats[0].cpu(), z.sum(dim=(0, 2)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(stats[1].cpu(), (z * z).sum(dim=(0, 2)), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("B,C,T,Cout,k,nsplit", [(3, 1024, 50, 192, 7, 2), (2, 1536, 64, 192, 7, 5), (1, 1088, 70, 64, 3, 8)])
def test_conv1d_fwd_splitk_equals_the_whole_launch(B, C, T, Cout, k, nsplit):
    """mm_conv1d_fwd_splitk (input channels reduced in `nsplit` slices by nsplit x the workgroups, slices added in order by
    a second launch that runs the epilogue) against mm_conv1d_fwd on the same operands: fp32 pre-BatchNorm output and the
    activation statistics to fp32 re-association level (1e-5 of the output scale), the bf16 pooled GELU output to one bf16
    step, both against F.conv1d; ragged last row tile, a last slice shorter than the others (17 chunks in 8 slices), run
    twice -> bit-identical (no atomics).  The plan function picks the config-#5 shape and leaves the others alone."""
    import ctypes
    hip = _hip()
    g = torch.Generator().manual_seed(B + C + T)
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(Cout, C, k, generator=g) / math.sqrt(C * k)
    bias = torch.randn(Cout, generator=g)
    xg = torch.empty(B, T, C, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pack_nct_bf16", x.cuda(), xg, B, C, T, C)
    wf, _ = _prep_w(hip, w, C)

    def run(split, pooled):
        stats = torch.zeros(32, 2, Cout, device="cuda")
        of = None if pooled else torch.empty(B, T, Cout, device="cuda")
        ob = torch.empty(B, T // 2, Cout, dtype=torch.bfloat16, device="cuda") if pooled else None
        args = (xg, wf, B, T, C, Cout, k, k // 2, None, bias.cuda(), 1 if pooled else 0, None, None, 2 if pooled else 1,
                stats, of, ob, None, 0.0, 0, None, None, 0)
        if split:
            ws = torch.full((split * B * T * Cout,), float("nan"), device="cuda")
            hip.call("mm_conv1d_fwd_splitk", *args, ws, split)
        else:
            hip.call("mm_conv1d_fwd", *args)
        return (ob if pooled else of), _stat(stats)
    z = F.conv1d(_bf(x), _bf(w), bias, padding=k // 2)
    for pooled in (False, True):
        if pooled and T % 2:
            continue
        y0, s0 = run(0, pooled)
        y1, s1 = run(nsplit, pooled)
        y2, s2 = run(nsplit, pooled)
        assert torch.equal(y1, y2) and torch.equal(s1, s2)
        if pooled:
            want = F.max_pool1d(F.gelu(z), 2).transpose(1, 2)
            torch.testing.assert_close(y1.float().cpu(), want, rtol=1e-2, atol=1e-2)
            assert (y1.float() - y0.float()).abs().max().item() <= 2 ** -7 * max(1.0, y0.float().abs().max().item())
        else:
            torch.testing.assert_close(y1.cpu(), z.transpose(1, 2), rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(s1, s0, rtol=1e-5, atol=1e-3)
    plan = {}
    for shape in ((32, 64, 6272, 192, 7), (32, 1024, 64, 128, 7), (32, 512, 128, 128, 1), (2, 64, 1536, 192, 7)):
        n, f = ctypes.c_int(0), ctypes.c_int64(0)
        hip.call("mm_conv1d_fwd_splitk_plan", *shape, ctypes.addressof(n), ctypes.addressof(f))
        plan[shape] = (n.value, f.value)
    assert plan[(32, 64, 6272, 192, 7)] == (5, 5 * 32 * 64 * 192)
    assert plan[(32, 1024, 64, 128, 7)] == (1, 0) and plan[(32, 512, 128, 128, 1)] == (1, 0)
    assert plan[(2, 64, 1536, 192, 7)][0] == 3
    with pytest.raises(hip.HipLibraryError):
        hip.call("mm_conv1d_fwd_splitk", xg, wf, B, T, C, Cout, k, k // 2, None, None, 0, None, None, 1, None,
                 torch.empty(B, T, Cout, device="cuda"), None, None, 0.0, 0, None, None, 0, None, nsplit)


def _attn_ref(qkv, H):
    B, L, E3 = qkv.shape
    E = E3 // 3
    dh = E // H
    q, k, v = (t.view(B, L, H, dh).transpose(1, 2) for t in qkv.split(E, dim=2))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, L, E)
    return o, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("B,L,H", [(2, 512, 4), (3, 125, 4), (1, 300, 2), (2, 32, 1)])
def test_attention_fwd_bwd(B, L, H):
    hip = _hip()
    g = torch.Generator().manual_seed(L)
    E = H * 32
    qkv = _bf(torch.randn(B, L, 3 * E, generator=g))
    do = _bf(torch.randn(B, L, E, generator=g))
    qg = qkv.cuda().to(torch.bfloat16)
    out = torch.empty(B, L, E, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, L, device="cuda")
    hip.call("mm_attn_fwd", qg, out, lse, B, L, H, 32, 1 / math.sqrt(32), 0.0, 0, None, None, 0)
    qr = qkv.clone().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, H)
    torch.testing.assert_close(out.float().cpu(), o_ref.detach(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(lse.cpu(), lse_ref.detach(), rtol=1e-3, atol=1e-3)
    o_ref.backward(do)
    dqkv = torch.empty_like(qg)
    delta = torch.empty(B, H, L, device="cuda")
    hip.call("mm_attn_bwd", qg, out, do.cuda().to(torch.bfloat16), lse, dqkv, delta, B, L, H, 32, 1 / math.sqrt(32), 0.0, 0, None, None, 0)
    got, want = dqkv.float().cpu(), qr.grad
    assert ((got - want).norm() / want.norm()).item() < 2e-2
    torch.testing.assert_close(got, want, rtol=5e-2, atol=3e-2)


@pytest.mark.parametrize("B,C,T,Cout,k", [(2, 64, 256, 64, 7), (3, 64, 200, 128, 5), (2, 128, 128, 128, 3),
                                          (4, 8, 256, 64, 7), (1, 128, 1000, 384, 1), (2, 512, 64, 128, 1),
                                          (6, 1024, 20, 768, 3), (5, 2048, 33, 192, 7)])
def test_conv1d_wgrad_matches_autograd(B, C, T, Cout, k):
    """(the last two shapes: short sequences with 192 / 96 output tiles - the config-#5 kind - where a workgroup accumulates
    over a GROUP of samples, so that the slot count does not grow with the batch: 2 and 3 slots instead of 6 and 5)"""
    hip = _hip()
    g = torch.Generator().manual_seed(C + T + k)
    x = _bf(torch.randn(B, C, T, generator=g))
    dy = _bf(torch.randn(B, Cout, T, generator=g))
    w = torch.zeros(Cout, C, k, requires_grad=True)
    bias = torch.zeros(Cout, requires_grad=True)
    F.conv1d(x, w, bias, padding=k // 2).backward(dy)
    cp = _cpad(C)
    xg = torch.empty(B, T, cp, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pack_nct_bf16", x.cuda(), xg, B, C, T, cp)
    dyg = dy.transpose(1, 2).contiguous().cuda().to(torch.bfloat16)
    # one slot per row-chunk workgroup, plain stores into UNINITIALISED memory; here with the strides of the PyTorch layout
    import ctypes
    n = ctypes.c_int(0)
    hip.call("mm_conv1d_wgrad_slots", B, T, cp, Cout, k, ctypes.addressof(n))
    slots = n.value
    assert slots >= 1
    if T <= 64 and C >= 1024:
        assert slots == {1024: 2, 2048: 3}[C], slots
    dw = torch.full((slots, Cout, C, k), float("nan"), device="cuda")
    db = torch.zeros(32, Cout, device="cuda")
    hip.call("mm_conv1d_wgrad", dyg, xg, dw, db, B, T, cp, Cout, k, k // 2, C, C * k, k, 1, slots, Cout * C * k, 1)
    dw, db = dw.sum(0), _grad(db)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(db.cpu(), bias.grad, rtol=2e-3, atol=2e-2)
    with pytest.raises(hip.HipLibraryError, match="slot_mode"):          # the fp32-atomics mode is gone
        hip.call("mm_conv1d_wgrad", dyg, xg, dw, db, B, T, cp, Cout, k, k // 2, C, C * k, k, 1, 2, Cout * C * k, 0)
    # channel-contiguous slots, summed (in slot order) and moved to the parameter layout by the scatter
    ws = torch.full((slots, Cout, k, cp), float("nan"), device="cuda")
    hip.call("mm_conv1d_wgrad", dyg, xg, ws, None, B, T, cp, Cout, k, k // 2, cp, k * cp, 1, cp, slots, Cout * k * cp, 1)
    dw2 = torch.zeros(Cout, C, k, device="cuda")
    hip.call("mm_wgrad_scatter", ws, dw2, Cout, C, k, cp, slots)
    torch.testing.assert_close(dw2.cpu(), w.grad, rtol=2e-3, atol=2e-2)


def test_conv1d_dgrad_via_forward_kernel():
    """data gradient = forward kernel on dY with the flipped/transposed image"""
    hip = _hip()
    g = torch.Generator().manual_seed(77)
    B, C, T, Cout, k = 2, 64, 128, 128, 5
    x = torch.randn(B, C, T, generator=g, requires_grad=True)
    w = _bf(torch.randn(Cout, C, k, generator=g) / 10)
    dy = _bf(torch.randn(B, Cout, T, generator=g))
    F.conv1d(x, w, None, padding=k // 2).backward(dy)
    _, wd = _prep_w(hip, w, C, Cout)
    dyg = dy.transpose(1, 2).contiguous().cuda().to(torch.bfloat16)
    dx = torch.empty(B, T, C, device="cuda")
    hip.call("mm_conv1d_fwd", dyg, wd, B, T, Cout, C, k, k - 1 - k // 2, None, None, 0, None, None, 1,
             None, dx, None, None, 0.0, 0, None, None, 0)
    torch.testing.assert_close(dx.cpu().transpose(1, 2), x.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("M,D", [(1000, 128), (37, 64), (64, 512)])
def test_layernorm_fwd_bwd(M, D):
    hip = _hip()
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, D, generator=g) * 2 + 0.5
    gam = 0.5 + torch.rand(D, generator=g)
    bet = torch.randn(D, generator=g)
    dy = _bf(torch.randn(M, D, generator=g))
    dres = torch.randn(M, D, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.layer_norm(xr, (D,), gr, br, 1e-5)
    y.backward(dy)
    out = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    stat = torch.empty(M, 2, device="cuda")
    hip.call("mm_layernorm_fwd", x.cuda(), gam.cuda(), bet.cuda(), out, None, stat, M, D, 1e-5)
    torch.testing.assert_close(out.float().cpu(), y.detach(), rtol=1e-2, atol=1e-2)
    dx = torch.empty(M, D, device="cuda")
    dgb = torch.zeros(32, 2, D, device="cuda")
    hip.call("mm_layernorm_bwd", dy.cuda().to(torch.bfloat16), None, x.cuda(), stat, gam.cuda(), dres.cuda(), dx, None,
             dgb, M, D, 0.0, 0, None)
    dg, db = _grad(dgb)[0], _grad(dgb)[1]
    torch.testing.assert_close(dx.cpu(), xr.grad + dres, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dg.cpu(), gr.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("N", [128, 16, 48, 100])
def test_clip_loss_own_rows_vs_oracle_and_bit_reproducible(N):
    """a-X2 / a-X5: mm_clip_loss_own_rows on rank r (four emulated ranks on one GPU) == autograd of the CPU
    oracle: scal = this rank's mean loss / top-1 / d loss_r / d logit_scale, dz = d (SUM over ranks of their
    losses) / d (its own embeddings) - the block a reduce-scatter-sum of every rank's gradient w.r.t. the
    gathered batch would deliver.  N not a multiple of 32 exercises the dot-product tail (round-1 bug: double
    counting).  No float atomics: two runs are bit-identical.  fp32 kernels: 1e-5."""
    hip = _hip()
    g = torch.Generator().manual_seed(21)
    W, B = 4, 16
    Bg = W * B
    z_cpu = torch.cat([F.normalize(torch.randn(Bg, N, generator=g), dim=1),
                       F.normalize(torch.randn(Bg, N, generator=g), dim=1)], dim=1)
    z_all = z_cpu.cuda().contiguous()
    ls0 = math.log(1 / 0.07)
    ls = torch.tensor([ls0], device="cuda")
    za = z_cpu.clone().requires_grad_(True)
    lso = torch.tensor(ls0, requires_grad=True)
    per_rank = []
    for r in range(W):
        zr = za[r * B:(r + 1) * B]
        per_rank.append(RF.clip_loss(zr[:, :N], za[:, N:], za[:, :N], zr[:, N:], lso.exp(), row0=r * B))
    dls = [torch.autograd.grad(pr[0], lso, retain_graph=True)[0] for pr in per_rank]
    sum(pr[0] for pr in per_rank).backward()
    ws = torch.empty(6 * Bg, device="cuda")
    for r in range(W):
        dz = torch.full((B, 2 * N), float("nan"), device="cuda")
        scal = torch.full((4,), float("nan"), device="cuda")
        hip.call("mm_clip_loss_own_rows", z_all, ls, scal, dz, ws, B, Bg, N, r * B)
        torch.testing.assert_close(dz.cpu(), za.grad[r * B:(r + 1) * B], rtol=1e-4, atol=1e-6)
        want = torch.stack([per_rank[r][0].detach(), per_rank[r][1], per_rank[r][2], dls[r]])
        torch.testing.assert_close(scal.cpu(), want, rtol=1e-5, atol=1e-6)
        dz2, scal2 = torch.empty_like(dz), torch.empty_like(scal)
        hip.call("mm_clip_loss_own_rows", z_all, ls, scal2, dz2, ws, B, Bg, N, r * B)
        assert torch.equal(dz, dz2) and torch.equal(scal, scal2)
    with pytest.raises(Exception):
        hip.call("mm_clip_loss_own_rows", z_all, ls, scal, dz, ws, B, Bg, N, Bg)
    with pytest.raises(Exception, match="multiple of 4"):
        hip.call("mm_clip_loss_own_rows", z_all, ls, scal, dz, ws, B, Bg, 6, 0)


def test_grouped_linear_wgrads_equal_separate_launches():
    """mm_conv1d_wgrad_many (one launch, workgroup id -> problem; fewer, longer row chunks per problem) sums to the
    same weight gradients and bias partials as one mm_conv1d_wgrad(slot_mode=1) launch per problem; 14 problems
    also exercise the split into two tables of 12."""
    import ctypes
    import struct
    hip = _hip()
    g = torch.Generator().manual_seed(5)
    shapes = [(512, 128, 384), (512, 512, 128), (96, 128, 128), (32, 128, 64)] + [(64, 48, 32)] * 10
    keep, descs, ref = [], [], []
    for M, cin, cout in shapes:
        dy = torch.randn(M, cout, generator=g).cuda().to(torch.bfloat16)
        x = torch.randn(M, cin, generator=g).cuda().to(torch.bfloat16)
        n, n1 = ctypes.c_int(0), ctypes.c_int(0)
        hip.call("mm_conv1d_wgrad_many_slots", 1, M, cin, cout, ctypes.addressof(n))
        hip.call("mm_conv1d_wgrad_slots", 1, M, cin, cout, 1, ctypes.addressof(n1))
        slots = n.value
        assert 1 <= slots <= n1.value
        ws_a = torch.full((n1.value, cout, cin), float("nan"), device="cuda")
        ws_b = torch.full((slots, cout, cin), float("nan"), device="cuda")
        db_a, db_b = torch.zeros(32, cout, device="cuda"), torch.zeros(32, cout, device="cuda")
        hip.call("mm_conv1d_wgrad", dy, x, ws_a, db_a, 1, M, cin, cout, 1, 0, cin, cin, 1, cin, n1.value, cout * cin, 1)
        descs.append(struct.pack("<QQQQiiiiiiii", dy.data_ptr(), x.data_ptr(), ws_b.data_ptr(), db_b.data_ptr(),
                                 1, M, cin, cout, cin, slots, 0, 0))
        keep.append((dy, x))
        ref.append((ws_a, ws_b, db_a, db_b, dy, x))
    raw = b"".join(descs)
    host = ctypes.create_string_buffer(raw, len(raw))
    hip.call("mm_conv1d_wgrad_many", ctypes.addressof(host), len(shapes))
    torch.cuda.synchronize()
    for ws_a, ws_b, db_a, db_b, dy, x in ref:
        assert not torch.isnan(ws_b).any()
        torch.testing.assert_close(ws_a.sum(0), ws_b.sum(0), rtol=1e-4, atol=1e-3)      # other chunking, same sums
        torch.testing.assert_close(_grad(db_a), _grad(db_b), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(_grad(db_b), dy.float().sum(0), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(ws_b.sum(0), dy.float().t() @ x.float(), rtol=2e-3, atol=2e-2)
    bad = struct.pack("<QQQQiiiiiiii", keep[0][0].data_ptr(), keep[0][1].data_ptr(), ref[0][1].data_ptr(), 0,
                      1, 512, 128, 384, 128, 1, 0, 0)                      # one slot is not enough
    hb = ctypes.create_string_buffer(bad, len(bad))
    with pytest.raises(Exception):
        hip.call("mm_conv1d_wgrad_many", ctypes.addressof(hb), 1)


def test_linear_with_fused_mean_over_time_and_pooled_head():
    """encoder tail: mm_linear_fwd_meanpool == mm_conv1d_fwd's rows + their per-group mean (a 64-bit
    fixed-point accumulator: 1e-5); mm_pooled_head_fwd / _bwd vs torch autograd of mean -> Linear -> GELU in fp32 (1e-5 / 1e-4; the saved
    pre-activation is bf16, so the backward's GELU' sees a rounded z: 1e-2 on d tokens)."""
    hip = _hip()
    g = torch.Generator().manual_seed(9)
    B, L, D, K, N = 4, 128, 128, 512, 96
    x = _bf(torch.randn(B * L, K, generator=g))
    w = _bf(torch.randn(D, K, generator=g) / math.sqrt(K))
    bias, res = torch.randn(D, generator=g), torch.randn(B * L, D, generator=g)
    wf, _ = _prep_w(hip, w.view(D, K, 1), K)
    xb = x.cuda().to(torch.bfloat16)
    out_a = torch.empty(B * L, D, device="cuda")
    out_b = torch.empty(B * L, D, device="cuda")
    pooled_acc = torch.zeros(B, 2 * D, device="cuda")         # B x D 64-bit elements
    hip.call("mm_conv1d_fwd", xb, wf, 1, B * L, K, D, 1, 0, None, bias.cuda(), 0, res.cuda(), None, 1, None, out_a,
             None, None, 0.0, 0, None, None, 0)
    hip.call("mm_linear_fwd_meanpool", xb, wf, B * L, K, bias.cuda(), res.cuda(), out_b, 0.0, 0, None, pooled_acc, L)
    assert torch.equal(out_a, out_b)
    pooled = (pooled_acc.view(torch.int64).double() * 2.0 ** -40).float()
    torch.testing.assert_close(out_b.cpu(), x @ w.t() + bias + res, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(pooled, out_b.view(B, L, D).mean(1), rtol=1e-5, atol=1e-5)
    with pytest.raises(Exception):
        hip.call("mm_linear_fwd_meanpool", xb, wf, B * L, K, None, None, out_b, 0.0, 0, None, pooled_acc, 96)
    # head forward / backward
    W = torch.randn(N, D, generator=g) / math.sqrt(D)
    hb = torch.randn(N, generator=g)
    tok = out_b.cpu().view(B, L, D).clone().requires_grad_(True)
    Wr, br = W.clone().requires_grad_(True), hb.clone().requires_grad_(True)
    y = F.gelu(tok.mean(1) @ Wr.t() + br)
    dout = torch.randn(B, N, generator=g)
    y.backward(dout)
    out = torch.empty(B, N, device="cuda")
    z = torch.empty(B, N, dtype=torch.bfloat16, device="cuda")
    pb = torch.empty(B, D, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pooled_head_fwd", pooled, None, W.cuda(), hb.cuda(), out, z, pb, B, D, N, 1, 0.0, 0, None)
    torch.testing.assert_close(out.cpu(), y.detach(), rtol=1e-4, atol=1e-5)
    out_acc = torch.empty(B, N, device="cuda")                # the same head, reading the accumulator itself
    hip.call("mm_pooled_head_fwd", None, pooled_acc, W.cuda(), hb.cuda(), out_acc, None, None, B, D, N, 1, 0.0, 0, None)
    assert torch.equal(out_acc, out)
    with pytest.raises(hip.HipLibraryError):
        hip.call("mm_pooled_head_fwd", pooled, pooled_acc, W.cuda(), hb.cuda(), out_acc, None, None, B, D, N, 1, 0.0, 0, None)
    torch.testing.assert_close(pb.float(), pooled, rtol=1e-2, atol=1e-2)
    dz = torch.empty(B, N, dtype=torch.bfloat16, device="cuda")
    dx = torch.empty(B, L, D, device="cuda")
    dxb = torch.empty(B, L, D, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pooled_head_bwd", dout.cuda(), z, W.cuda(), dz, dx, dxb, B, L, D, N, 1, 0.0, 0, 0.0, 0, None)
    torch.testing.assert_close(dx.cpu(), tok.grad, rtol=1e-2, atol=1e-2 * tok.grad.abs().max().item())
    torch.testing.assert_close(dxb.float(), dx, rtol=1e-2, atol=1e-6)
    torch.testing.assert_close(dz.float().cpu().t() @ pb.float().cpu(), Wr.grad, rtol=2e-2, atol=2e-2 * Wr.grad.abs().max().item())
    # dropout on the head output and on the emitted operand: same masks as mm_act_f32 / mm_act_bwd would draw
    hip.call("mm_pooled_head_bwd", dout.cuda(), z, W.cuda(), None, dx, dxb, B, L, D, N, 1, 0.0, 0, 0.4, 123, None)
    kept = dxb.float() != 0
    assert abs(kept.float().mean().item() - 0.6) < 0.02
    torch.testing.assert_close(dxb.float()[kept], (dx / 0.6)[kept], rtol=1e-2, atol=1e-6)
    ref_mask = torch.empty(B * L * D, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_act_bwd", torch.ones(B * L * D, device="cuda"), None, None, ref_mask, B * L * D, 0, 0.4, 123, None)
    assert torch.equal(ref_mask.view(B, L, D) != 0, kept | (dx == 0))


def test_compiled_in_epilogues_equal_the_generic_one_bit_for_bit(monkeypatch):
    """csrc/igemm1d.hip compiles the conv1d / Linear epilogue once per feature combination of the training step
    (launch_fwd's EPI_CASE table) besides the generic run-time form.  Every combination reachable through
    mm_conv1d_fwd at a shape that selects its tile is run both ways (MM_EPI_GENERIC=1 forces the generic kernel)
    and must agree bit for bit - outputs, pre-activation copy, BatchNorm sums."""
    hip = _hip()
    g = torch.Generator().manual_seed(5)
    M = 16384

    def run(K, N, taps, **kw):
        B, T = (32, 512) if taps > 1 else (1, M)
        x = _bf(torch.randn(B, T, K, generator=g)).cuda().to(torch.bfloat16)
        w = _bf(torch.randn(N, K, taps, generator=g) / math.sqrt(K * taps))
        wf, _ = _prep_w(hip, w, K)
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn(B, T, N, generator=g).cuda() if kw.get("res") else None
        gz = _bf(torch.randn(B, T, N, generator=g)).cuda().to(torch.bfloat16) if kw.get("gradz") else None
        outs = []
        for generic in (False, True):
            if generic:
                monkeypatch.setenv("MM_EPI_GENERIC", "1")
            else:
                monkeypatch.delenv("MM_EPI_GENERIC", raising=False)
            of = torch.full((B, T, N), float("nan"), device="cuda") if kw.get("f32") else None
            ob = None if kw.get("f32") else torch.full((B, T, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            op = torch.full((B, T, N), float("nan"), dtype=torch.bfloat16, device="cuda") if kw.get("pre") else None
            st = torch.zeros(32, 2, N, device="cuda") if kw.get("stats") else None
            hip.call("mm_conv1d_fwd", x, wf, B, T, K, N, taps, taps // 2, None, None if kw.get("nobias") else bias,
                     1 if kw.get("gelu") else 0, res, None, 1, st, of, ob, op, kw.get("p", 0.0), 17, None, gz, 1 if gz is not None else 0)
            torch.cuda.synchronize()
            outs.append([t.clone() for t in (of, ob, op, st) if t is not None])
        monkeypatch.delenv("MM_EPI_GENERIC", raising=False)
        for a_, b_ in zip(*outs):
            if a_.dtype == torch.float32 and a_.shape[0] == 32 and a_.dim() == 3 and a_.shape[1] == 2:
                assert torch.equal(_stat(a_), _stat(b_))       # fixed-point statistics: order-free, so bit-equal too
            else:
                assert torch.equal(a_, b_)

    run(128, 384, 1)                                        # QKV projection
    run(128, 512, 1, gelu=True, p=0.1, pre=True)            # FFN-1 forward
    run(128, 512, 1, nobias=True, p=0.1, gradz=True)        # FFN-2 data gradient
    run(128, 128, 1, nobias=True)                           # plain data gradient (32-row tile)
    run(64, 64, 7, f32=True, stats=True)                    # conv block forward
    run(128, 64, 5, nobias=True)                            # conv data gradient


@pytest.mark.parametrize("M,K,p", [(512, 128, 0.0), (96, 512, 0.2)])
def test_linear_with_fused_next_layernorm(M, K, p):
    """mm_linear_fwd_ln == mm_conv1d_fwd (same rows, bit for bit, same dropout mask) followed by
    mm_layernorm_fwd on them (bf16 rows equal up to one bf16 ulp, statistics 1e-5)."""
    hip = _hip()
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    w = _bf(torch.randn(128, K, generator=g) / math.sqrt(K))
    bias, res = torch.randn(128, generator=g).cuda(), (torch.randn(M, 128, generator=g) * 2).cuda()
    gam, bet = (0.5 + torch.rand(128, generator=g)).cuda(), torch.randn(128, generator=g).cuda()
    wf, _ = _prep_w(hip, w.view(128, K, 1), K)
    ref = torch.empty(M, 128, device="cuda")
    hip.call("mm_conv1d_fwd", x, wf, 1, M, K, 128, 1, 0, None, bias, 0, res, None, 1, None, ref, None, None, p, 11, None,
             None, 0)
    h_ref = torch.empty(M, 128, dtype=torch.bfloat16, device="cuda")
    st_ref = torch.empty(M, 2, device="cuda")
    hip.call("mm_layernorm_fwd", ref, gam, bet, h_ref, None, st_ref, M, 128, 1e-5)
    out = torch.empty(M, 128, device="cuda")
    h = torch.empty(M, 128, dtype=torch.bfloat16, device="cuda")
    st = torch.empty(M, 2, device="cuda")
    hip.call("mm_linear_fwd_ln", x, wf, M, K, bias, res, out, p, 11, None, gam, bet, 1e-5, h, st)
    assert torch.equal(out, ref)
    torch.testing.assert_close(st, st_ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(h.float(), h_ref.float(), rtol=8e-3, atol=1e-2)
    assert (h.float() - h_ref.float()).abs().gt(0).float().mean().item() < 0.02     # a handful of 1-ulp flips at most
    with pytest.raises(Exception):
        hip.call("mm_linear_fwd_ln", x, wf, M - 1, K, bias, res, out, p, 11, None, gam, bet, 1e-5, h, st)


@pytest.mark.parametrize("M,K", [(512, 384), (96, 512), (64, 48)])
def test_linear_dgrad_fused_with_layernorm_backward(M, K):
    """mm_linear_dgrad_ln_bwd = data gradient of Linear(128 -> K) followed by the LayerNorm-128 backward
    (+ skip-path gradient), vs torch autograd of Linear(LayerNorm(x)) in fp32 with the same bf16-rounded
    operands: dx 2e-3 (bf16 MFMA operands, fp32 accumulate), dgamma / dbeta 2e-3 relative to their scale."""
    hip = _hip()
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, 128, generator=g) * 2 + 0.5
    gam = 0.5 + torch.rand(128, generator=g)
    bet = torch.randn(128, generator=g)
    w = _bf(torch.randn(K, 128, generator=g) * 0.1)
    dy = _bf(torch.randn(M, K, generator=g))
    dres = torch.randn(M, 128, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    (F.layer_norm(xr, (128,), gr, br, 1e-5) @ w.t()).backward(dy)
    out = torch.empty(M, 128, dtype=torch.bfloat16, device="cuda")
    stat = torch.empty(M, 2, device="cuda")
    hip.call("mm_layernorm_fwd", x.cuda(), gam.cuda(), bet.cuda(), out, None, stat, M, 128, 1e-5)
    _, wd = _prep_w(hip, w.view(K, 128, 1), 128, K)
    dx = torch.empty(M, 128, device="cuda")
    dxb = torch.empty(M, 128, dtype=torch.bfloat16, device="cuda")
    dgb = torch.zeros(32, 2, 128, device="cuda")
    hip.call("mm_linear_dgrad_ln_bwd", dy.cuda().to(torch.bfloat16), wd, M, K, x.cuda(), stat, gam.cuda(), dres.cuda(),
             dx, dxb, dgb, 0.0, 0, None)
    torch.testing.assert_close(dx.cpu(), xr.grad + dres, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(dxb.float().cpu(), dx.cpu(), rtol=1e-2, atol=1e-2)
    dg, db = _grad(dgb)[0].cpu(), _grad(dgb)[1].cpu()
    torch.testing.assert_close(dg, gr.grad, rtol=2e-3, atol=2e-3 * gr.grad.abs().max().item())
    torch.testing.assert_close(db, br.grad, rtol=2e-3, atol=2e-3 * br.grad.abs().max().item())
    # dropout on the bf16 copy only: kept elements are dx / (1 - p), the fp32 output is untouched
    dx2 = torch.empty_like(dx)
    hip.call("mm_linear_dgrad_ln_bwd", dy.cuda().to(torch.bfloat16), wd, M, K, x.cuda(), stat, gam.cuda(), dres.cuda(),
             dx2, dxb, None, 0.25, 77, None)
    assert torch.equal(dx2, dx)
    kept = dxb.float() != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.03
    torch.testing.assert_close(dxb.float()[kept], (dx / 0.75)[kept], rtol=1e-2, atol=1e-2)
    with pytest.raises(Exception):
        hip.call("mm_linear_dgrad_ln_bwd", dy.cuda().to(torch.bfloat16), wd, M - 1, K, x.cuda(), stat, gam.cuda(),
                 None, dx, None, None, 0.0, 0, None)


@pytest.mark.parametrize("pool", [1, 2])
@pytest.mark.parametrize("N", [128, 48])
def test_bn_act_pool_train_fwd_bwd(pool, N):
    hip = _hip()
    g = torch.Generator().manual_seed(N + pool)
    R, S = 3, 64
    y = torch.randn(R, S, N, generator=g) * 1.5 + 0.3
    gam = 0.5 + torch.rand(N, generator=g)
    bet = torch.randn(N, generator=g) * 0.2
    rm, rv = torch.zeros(N), torch.ones(N)
    yr, gr, br = y.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    z = F.batch_norm(yr.permute(0, 2, 1), rm.clone(), rv.clone(), gr, br, training=True, momentum=0.1, eps=1e-5)
    a = F.gelu(z)
    if pool == 2:
        a = F.max_pool1d(a, 2)
    a = a.permute(0, 2, 1)
    dout = _bf(torch.randn(R, S // pool, N, generator=g))
    a.backward(dout)
    yg = y.cuda()
    from multimodal_eeg_fmri_amd.ops import ACC_STAT, acc_encode
    stats = acc_encode(torch.stack([y.sum(dim=(0, 1)), (y * y).sum(dim=(0, 1))]), ACC_STAT).cuda()
    rmg, rvg = rm.cuda(), rv.cuda()
    out4 = torch.empty(4, N, device="cuda")
    nbt = torch.full((), 7, dtype=torch.int64, device="cuda")
    hip.call("mm_bn_finalize", stats, gam.cuda(), bet.cuda(), rmg, rvg, None, out4, N, float(R * S), 0.1, 1e-5, 0, nbt)
    assert nbt.item() == 8                      # num_batches_tracked += 1, as nn.BatchNorm in train mode
    rm_ref, rv_ref = torch.zeros(N), torch.ones(N)
    F.batch_norm(y.permute(0, 2, 1), rm_ref, rv_ref, gam, bet, training=True, momentum=0.1, eps=1e-5)
    torch.testing.assert_close(rmg.cpu(), rm_ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rvg.cpu(), rv_ref, rtol=1e-4, atol=1e-5)
    ob = torch.empty(R, S // pool, N, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_bn_act_fwd", yg, out4[0], out4[1], None, ob, None, R, S, N, 1, pool, 1, 0.0, 0, 0.0, 0, None)
    torch.testing.assert_close(ob.float().cpu(), a.detach(), rtol=1e-2, atol=1e-2)
    sums = torch.zeros(32, 2, N, device="cuda")
    dg = dout.cuda().to(torch.bfloat16)
    hip.call("mm_bn_act_bwd_reduce", yg, out4, dg, None, sums, R, S, N, 1, pool, 1, 0.0, 0, 0.0, 0, None)
    torch.testing.assert_close(_grad(sums)[0].cpu(), br.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(_grad(sums)[1].cpu(), gr.grad, rtol=1e-3, atol=1e-3)
    dy = torch.empty(R, S, N, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_bn_act_bwd_apply", yg, out4, dg, None, _grad(sums).contiguous(), dy, None, R, S, N, 1, pool, 1, 0.0, 0, 0.0, 0, None, 1, 1)
    torch.testing.assert_close(dy.float().cpu(), yr.grad, rtol=2e-2, atol=2e-3)
    dy2 = torch.empty_like(dy)                      # same, the kernel summing the workspace's replicas itself
    hip.call("mm_bn_act_bwd_apply", yg, out4, dg, None, sums, dy2, None, R, S, N, 1, pool, 1, 0.0, 0, 0.0, 0, None, 1, 32)
    torch.testing.assert_close(dy2.float(), dy.float(), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("B,T,Cin,Cout,k,pool,p,generic", [
    (32, 512, 128, 128, 3, 2, 0.3, False),       # EEG conv block 3's data gradient over block 2 (MaxPool1d(2) between), C2 size
    (32, 1024, 128, 64, 5, 1, 0.3, False),       # block 2's over block 1
    (3, 100, 64, 48, 7, 1, 0.0, False),          # ragged row tile, ragged column tile, no dropout
    (2, 70, 32, 64, 3, 2, 0.2, True),            # generic (run-time) epilogue
])
def test_conv1d_dgrad_with_fused_bn_backward_reduce(B, T, Cin, Cout, k, pool, p, generic, monkeypatch):
    """mm_conv1d_dgrad_bn_reduce = mm_conv1d_fwd (data gradient) + mm_bn_act_bwd_reduce of the block below:
    the same bf16 d(out), the same sums (up to fp32 partial-sum grouping)."""
    hip = _hip()
    if generic:
        monkeypatch.setenv("MM_EPI_GENERIC", "1")
    g = torch.Generator().manual_seed(B * T + Cout)
    w = torch.randn(Cin, Cout, k, generator=g) / math.sqrt(Cout * k)       # the upper block's weight: (its Cout = Cin here, its Cin = Cout here)
    _, wd = _prep_w(hip, w, Cout, Cin)
    dy = (torch.randn(B, T, Cin, generator=g) * 0.1).cuda().to(torch.bfloat16)
    yb = (torch.randn(B, T * pool, Cout, generator=g) * 1.2 + 0.1).cuda()
    out4 = torch.stack([0.5 + torch.rand(Cout, generator=g), torch.randn(Cout, generator=g) * 0.2,
                        torch.randn(Cout, generator=g) * 0.1, 0.8 + 0.4 * torch.rand(Cout, generator=g)]).cuda().contiguous()
    seed = 4242
    dx_a = torch.full((B, T, Cout), float("nan"), device="cuda").to(torch.bfloat16)
    hip.call("mm_conv1d_fwd", dy, wd, B, T, Cin, Cout, k, k - 1 - k // 2, None, None, 0, None, None, 1,
             None, None, dx_a, None, 0.0, 0, None, None, 0)
    sums_a = torch.zeros(32, 2, Cout, device="cuda")
    hip.call("mm_bn_act_bwd_reduce", yb, out4, dx_a, None, sums_a, B, T * pool, Cout, 1, pool, 0, p, seed, 0.0, 0, None)
    dx_b = torch.full((B, T, Cout), float("nan"), device="cuda").to(torch.bfloat16)
    sums_b = torch.zeros(32, 2, Cout, device="cuda")
    hip.call("mm_conv1d_dgrad_bn_reduce", dy, wd, B, T, Cin, Cout, k, k - 1 - k // 2, dx_b, yb, out4, sums_b,
             1, pool, 0, p, seed, None)
    assert torch.equal(dx_a, dx_b) and torch.isfinite(dx_b.float()).all()
    sa, sb = _grad(sums_a).cpu(), _grad(sums_b).cpu()
    assert sa.abs().max() > 1e-2
    torch.testing.assert_close(sb, sa, rtol=1e-4, atol=1e-4 * float(sa.abs().max()))
    # the wide-Linear tiles have no such epilogue: refused, not ignored
    with pytest.raises(hip.HipLibraryError):
        hip.call("mm_conv1d_dgrad_bn_reduce", dy, wd, B, T, Cin, 128, 1, 0, dx_b, yb, out4, sums_b, 1, pool, 0, p, seed, None)


@pytest.mark.parametrize("B,D,H,W", [(32, 32, 32, 32), (3, 5, 7, 12), (2, 6, 4, 48), (2, 4, 6, 10), (1, 1, 1, 4)])
def test_conv3d_l1_tap_sums(B, D, H, W):
    """mm_conv3d_l1_tapsum: S[tap] = sum over output voxels of the zero-padded, bf16-rounded input at that tap
    (eight lanes per row when W % 4 == 0, a thread per row otherwise)."""
    hip = _hip()
    g = torch.Generator().manual_seed(B * 1000 + W)
    x = torch.randn(B, D, H, W, generator=g) + 0.3
    xp = F.pad(_bf(x).double(), (1, 1, 1, 1, 1, 1))
    ref = torch.stack([xp[:, kd:kd + D, kh:kh + H, kw:kw + W].sum() for kd in range(3) for kh in range(3) for kw in range(3)])
    ws = torch.zeros(32, 32, device="cuda")
    hip.call("mm_conv3d_l1_tapsum", x.cuda(), ws, B, D, H, W)
    got = _stat(ws)[:27].cpu().double()
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()) + 1e-4)


@pytest.mark.parametrize("B,D,H,W", [(4, 8, 16, 32), (3, 6, 10, 12), (2, 4, 8, 48), (32, 32, 32, 32)])
def test_conv3d_l1_gram_matrix_and_the_statistics_derived_from_it(B, D, H, W):
    """csrc/conv3d_l1.hip, training forward of the first voxel layer: G = Xcol^T Xcol (27 taps + a column of ones) of the
    zero-padded, bf16-rounded volume against the same matrix formed in fp64 on the CPU (full tiles, ragged H and W tiles,
    the C2 size); the BatchNorm sums every workgroup derives from its share of G (sum y = w . S + M b, sum y^2 =
    w^T G w + 2 b w . S + M b^2) against (a) the recompute pass of the same library (mm_conv3d_l1 mode 0) and (b)
    F.conv3d in fp64."""
    hip = _hip()
    g = torch.Generator().manual_seed(B * 100 + W)
    x = torch.randn(B, D, H, W, generator=g) + 0.3
    xp = F.pad(_bf(x).double(), (1, 1, 1, 1, 1, 1))
    cols = [xp[:, kd:kd + D, kh:kh + H, kw:kw + W].reshape(-1) for kd in range(3) for kh in range(3) for kw in range(3)]
    cols = torch.stack(cols + [torch.ones_like(cols[0])])                         # (28, M)
    Gref = cols @ cols.t()
    w = _bf(torch.randn(32, 27, generator=g) * 0.25)
    bias = torch.randn(32, generator=g) * 0.2
    wimg, _ = _prep_w(hip, w.reshape(32, 27, 1), 32)
    gram = torch.zeros(32, 32, 32, device="cuda")
    stats = torch.zeros(32, 2, 32, device="cuda")
    hip.call("mm_conv3d_l1_gram", x.cuda(), wimg, bias.cuda(), gram, stats, B, D, H, W)
    G = _stat(gram).cpu().double()
    assert torch.equal(G[28:], torch.zeros_like(G[28:])) and torch.equal(G[:, 28:], torch.zeros_like(G[:, 28:]))
    assert torch.equal(G.tril(-1), torch.zeros_like(G))                         # symmetric: only the upper triangle is accumulated
    G = G + G.triu(1).t()
    torch.testing.assert_close(G[:28, :28], Gref, rtol=3e-5, atol=3e-5 * float(Gref.abs().max()))
    assert G[27, 27].item() == B * D * H * W
    got = _stat(stats).cpu().double()
    y = F.conv3d(_bf(x).double().unsqueeze(1), w.double().view(32, 1, 3, 3, 3), bias.double(), padding=1)
    want = torch.stack([y.sum(dim=(0, 2, 3, 4)), (y * y).sum(dim=(0, 2, 3, 4))])
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
    stats0 = torch.zeros(32, 2, 32, device="cuda")
    hip.call("mm_conv3d_l1", 0, x.cuda(), wimg, bias.cuda(), None, None, None, stats0, None, None, None, B, D, H, W, 1, 0.0, 0, None)
    torch.testing.assert_close(got, _stat(stats0).cpu().double(), rtol=2e-5, atol=2e-5 * float(want.abs().max()))


def _bn_fin(stats, gam, bet, rm, rv, out4, nb, count, mom=0.1, eps=1e-5):
    """host-side mm_bn_fin_t (include/mmeeg_hip.h) -> (address, keep-alive)"""
    import ctypes
    import struct
    raw = struct.pack("<QQQQQQQfffi", stats.data_ptr(), gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                      out4.data_ptr(), nb.data_ptr() if nb is not None else 0, float(count), mom, eps, 0)
    buf = ctypes.create_string_buffer(raw, len(raw))
    return ctypes.addressof(buf), buf


@pytest.mark.parametrize("R,S,N,pool,ln", [(32, 512, 128, 1, True), (32, 1024, 128, 2, False), (32, 1024, 64, 1, False), (3, 38, 64, 2, False),
                                           (2, 100, 256, 1, False)])
def test_batchnorm_finalize_folded_into_the_apply_pass_is_bit_identical(R, S, N, pool, ln):
    """mm_bn_act_fwd_fin / mm_bn_act_fwd_ln_fin = mm_bn_finalize (train mode) + mm_bn_act_fwd[_ln]: the finalize runs in the
    apply pass's prologue (one launch, one graph node less per BatchNorm layer) with the SAME device function - outputs,
    out4, running statistics and num_batches_tracked equal bit for bit; an incomplete descriptor is refused."""
    hip = _hip()
    from multimodal_eeg_fmri_amd.ops import ACC_STAT, acc_encode
    g = torch.Generator().manual_seed(R * S + N)
    y = (torch.randn(R, S, N, generator=g) * 1.3 + 0.2).cuda()
    cnt = R * S
    stats = acc_encode(torch.stack([y.sum(dim=(0, 1)), (y * y).sum(dim=(0, 1))]), ACC_STAT).contiguous()
    gam, bet = (0.5 + torch.rand(N, generator=g)).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    pe = torch.randn(S // pool, N, generator=g).cuda()
    lg, lb = (0.5 + torch.rand(N, generator=g)).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    res = []
    for fused in (False, True):
        rm, rv = torch.full((N,), 0.25, device="cuda"), torch.full((N,), 1.5, device="cuda")
        nb = torch.zeros((), dtype=torch.long, device="cuda")
        out4 = torch.full((4, N), float("nan"), device="cuda")
        of = torch.full((R, S // pool, N), float("nan"), device="cuda")
        ob = None if ln else torch.full((R, S // pool, N), float("nan"), device="cuda").to(torch.bfloat16)
        hn = torch.full((R * S, N), float("nan"), device="cuda").to(torch.bfloat16) if ln else None
        st = torch.full((R * S, 2), float("nan"), device="cuda") if ln else None
        if fused:
            addr, keep = _bn_fin(stats, gam, bet, rm, rv, out4, nb, cnt)
            if ln:
                hip.call("mm_bn_act_fwd_ln_fin", y, addr, pe, of, R, S, 1, 0.3, 11, 0.1, 22, None, lg, lb, 1e-5, hn, st)
            else:
                hip.call("mm_bn_act_fwd_fin", y, addr, pe, ob, of, R, S, N, 1, pool, 0, 0.3, 11, 0.1, 22, None)
        else:
            hip.call("mm_bn_finalize", stats, gam, bet, rm, rv, None, out4, N, float(cnt), 0.1, 1e-5, 0, nb)
            if ln:
                hip.call("mm_bn_act_fwd_ln", y, out4[0], out4[1], pe, of, R, S, 1, 0.3, 11, 0.1, 22, None, lg, lb, 1e-5, hn, st)
            else:
                hip.call("mm_bn_act_fwd", y, out4[0], out4[1], pe, ob, of, R, S, N, 1, pool, 0, 0.3, 11, 0.1, 22, None)
        torch.cuda.synchronize()
        res.append([t.clone() for t in (of, out4, rm, rv, nb) + ((hn, st) if ln else (ob,))])
    for a, b in zip(*res):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    assert res[1][4].item() == 1
    mean = y.double().mean(dim=(0, 1)).cpu()
    torch.testing.assert_close(res[1][1][2].cpu().double(), mean, rtol=1e-4, atol=1e-5)
    bad = _bn_fin(stats, gam, bet, rm, rv, out4, nb, 0.0)
    with pytest.raises(hip.HipLibraryError):
        hip.call("mm_bn_act_fwd_fin", y, bad[0], pe, ob, of, R, S, N, 1, pool, 0, 0.0, 0, 0.0, 0, None)


def test_batchnorm_finalize_folded_into_the_voxel_apply_passes_is_bit_identical():
    """the same for the two voxel-encoder consumers: mm_pool3d_bn_act_fwd_fin (layer 2: bf16 pre-BatchNorm volume, pooled)
    and mm_conv3d_l1_fwd_fin (the fused first layer's forward kernel)."""
    hip = _hip()
    from multimodal_eeg_fmri_amd.ops import ACC_STAT, acc_encode
    g = torch.Generator().manual_seed(77)
    B, D, H, W, N = 4, 8, 8, 8, 64
    y = (torch.randn(B, D, H, W, N, generator=g) * 1.2 + 0.1).cuda().to(torch.bfloat16)
    yf = y.float()
    cnt = B * D * H * W
    stats = acc_encode(torch.stack([yf.sum(dim=(0, 1, 2, 3)), (yf * yf).sum(dim=(0, 1, 2, 3))]), ACC_STAT).contiguous()
    gam, bet = (0.5 + torch.rand(N, generator=g)).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    res = []
    for fused in (False, True):
        rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
        nb = torch.zeros((), dtype=torch.long, device="cuda")
        out4 = torch.full((4, N), float("nan"), device="cuda")
        out = torch.full((B, D // 2, H // 2, W // 2, N), float("nan"), device="cuda").to(torch.bfloat16)
        ysel, arg = torch.zeros_like(out), torch.zeros(out.shape, dtype=torch.uint8, device="cuda")
        if fused:
            addr, keep = _bn_fin(stats, gam, bet, rm, rv, out4, nb, cnt)
            hip.call("mm_pool3d_bn_act_fwd_fin", y, addr, out, ysel, arg, B, D, H, W, N, 1, 0.3, 5, None)
        else:
            hip.call("mm_bn_finalize", stats, gam, bet, rm, rv, None, out4, N, float(cnt), 0.1, 1e-5, 0, nb)
            hip.call("mm_pool3d_bn_act_fwd", y, out4, out, ysel, arg, B, D, H, W, N, 1, 0.3, 5, None)
        torch.cuda.synchronize()
        res.append([t.clone() for t in (out, ysel, arg, out4, rm, rv, nb)])
    assert all(torch.equal(a, b) for a, b in zip(*res)) and torch.isfinite(res[1][0].float()).all()
    # layer 1: Gram statistics -> [finalize ->] forward
    B, D, H, W = 3, 6, 8, 32
    x = (torch.randn(B, D, H, W, generator=g) + 0.2).cuda()
    w = _bf(torch.randn(32, 27, generator=g) * 0.25)
    bias = (torch.randn(32, generator=g) * 0.2).cuda()
    wimg, _ = _prep_w(hip, w.reshape(32, 27, 1), 32)
    gam, bet = (0.5 + torch.rand(32, generator=g)).cuda(), (torch.randn(32, generator=g) * 0.1).cuda()
    res = []
    for fused in (False, True):
        gram, stats = torch.zeros(32, 32, 32, device="cuda"), torch.zeros(32, 2, 32, device="cuda")
        hip.call("mm_conv3d_l1_gram", x, wimg, bias, gram, stats, B, D, H, W)
        rm, rv = torch.zeros(32, device="cuda"), torch.ones(32, device="cuda")
        nb = torch.zeros((), dtype=torch.long, device="cuda")
        out4 = torch.full((4, 32), float("nan"), device="cuda")
        out = torch.full((B, D // 2, H // 2, W // 2, 32), float("nan"), device="cuda").to(torch.bfloat16)
        if fused:
            addr, keep = _bn_fin(stats, gam, bet, rm, rv, out4, nb, B * D * H * W)
            hip.call("mm_conv3d_l1_fwd_fin", x, wimg, bias, addr, out, B, D, H, W, 0.2, 9, None)
        else:
            hip.call("mm_bn_finalize", stats, gam, bet, rm, rv, None, out4, 32, float(B * D * H * W), 0.1, 1e-5, 0, nb)
            hip.call("mm_conv3d_l1", 1, x, wimg, bias, out4, None, None, None, out, None, None, B, D, H, W, 1, 0.2, 9, None)
        torch.cuda.synchronize()
        res.append([t.clone() for t in (out, out4, rm, rv, nb)])
    assert all(torch.equal(a, b) for a, b in zip(*res)) and torch.isfinite(res[1][0].float()).all()


@pytest.mark.parametrize("R,S,p,p2", [(32, 512, 0.3, 0.1), (3, 37, 0.0, 0.0)])
def test_bn_act_with_fused_first_layernorm(R, S, p, p2):
    """mm_bn_act_fwd_ln = mm_bn_act_fwd (fp32 out, positional add, both dropouts) + mm_layernorm_fwd of its rows"""
    hip = _hip()
    N = 128
    g = torch.Generator().manual_seed(R * S)
    y = (torch.randn(R, S, N, generator=g) * 1.3 + 0.2).cuda()
    sc, sh = (0.5 + torch.rand(N, generator=g)).cuda(), (torch.randn(N, generator=g) * 0.2).cuda()
    pe = torch.randn(S, N, generator=g).cuda()
    gam, bet = (0.5 + torch.rand(N, generator=g)).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    o_a = torch.full((R, S, N), float("nan"), device="cuda")
    hip.call("mm_bn_act_fwd", y, sc, sh, pe, None, o_a, R, S, N, 1, 1, 1, p, 11, p2, 22, None)
    h_a = torch.empty(R * S, N, dtype=torch.bfloat16, device="cuda")
    st_a = torch.empty(R * S, 2, device="cuda")
    hip.call("mm_layernorm_fwd", o_a.view(R * S, N), gam, bet, h_a, None, st_a, R * S, N, 1e-5)
    o_b = torch.full((R, S, N), float("nan"), device="cuda")
    h_b = torch.full((R * S, N), float("nan"), device="cuda").to(torch.bfloat16)
    st_b = torch.full((R * S, 2), float("nan"), device="cuda")
    hip.call("mm_bn_act_fwd_ln", y, sc, sh, pe, o_b, R, S, 1, p, 11, p2, 22, None, gam, bet, 1e-5, h_b, st_b)
    assert torch.equal(o_a, o_b)
    torch.testing.assert_close(st_b, st_a, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(h_b.float(), h_a.float(), rtol=1e-2, atol=1e-2)
    ref = F.layer_norm(o_a.view(R * S, N).cpu(), (N,), gam.cpu(), bet.cpu(), 1e-5)
    torch.testing.assert_close(h_b.float().cpu(), ref, rtol=1e-2, atol=2e-2)


@pytest.mark.parametrize("M,K,p,p2,generic", [(16384, 384, 0.3, 0.1, False), (96, 384, 0.0, 0.0, False), (64, 128, 0.2, 0.3, True)])
def test_linear_dgrad_ln_backward_with_fused_bn_backward_reduce(M, K, p, p2, generic, monkeypatch):
    """mm_linear_dgrad_ln_bwd_bn_reduce = mm_linear_dgrad_ln_bwd (fp32 rows) + mm_bn_act_bwd_reduce of the 128-channel
    conv block those rows are the d(out) of (incl. the positional-encoding dropout in front of it)."""
    hip = _hip()
    if generic:
        monkeypatch.setenv("MM_EPI_GENERIC", "1")
    g = torch.Generator().manual_seed(M + K)
    w = torch.randn(K, 128, 1, generator=g) / math.sqrt(128)                # the Linear 128 -> K whose input is LN(x)
    _, wd = _prep_w(hip, w, 128, K)
    dy = (torch.randn(M, K, generator=g) * 0.1).cuda().to(torch.bfloat16)
    x = torch.randn(M, 128, generator=g).cuda()
    stat = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    gam = (0.5 + torch.rand(128, generator=g)).cuda()
    dres = (torch.randn(M, 128, generator=g) * 0.1).cuda()
    yb = (torch.randn(M, 128, generator=g) * 1.2 + 0.1).cuda()
    out4 = torch.stack([0.5 + torch.rand(128, generator=g), torch.randn(128, generator=g) * 0.2,
                        torch.randn(128, generator=g) * 0.1, 0.8 + 0.4 * torch.rand(128, generator=g)]).cuda().contiguous()
    dx_a = torch.full((M, 128), float("nan"), device="cuda")
    dgb_a = torch.zeros(32, 2, 128, device="cuda")
    hip.call("mm_linear_dgrad_ln_bwd", dy, wd, M, K, x, stat, gam, dres, dx_a, None, dgb_a, 0.0, 0, None)
    sums_a = torch.zeros(32, 2, 128, device="cuda")
    hip.call("mm_bn_act_bwd_reduce", yb, out4, None, dx_a, sums_a, 1, M, 128, 1, 1, 1, p, 31, p2, 32, None)
    dx_b = torch.full((M, 128), float("nan"), device="cuda")
    dgb_b = torch.zeros(32, 2, 128, device="cuda")
    sums_b = torch.zeros(32, 2, 128, device="cuda")
    hip.call("mm_linear_dgrad_ln_bwd_bn_reduce", dy, wd, M, K, x, stat, gam, dres, dx_b, dgb_b, None, yb, out4, sums_b,
             1, p, 31, p2, 32)
    assert torch.equal(dx_a, dx_b) and torch.isfinite(dx_b).all()
    assert torch.equal(dgb_a.view(torch.int32), dgb_b.view(torch.int32))
    sa, sb = _grad(sums_a).cpu(), _grad(sums_b).cpu()
    assert sa.abs().max() > 1e-3
    torch.testing.assert_close(sb, sa, rtol=1e-4, atol=1e-4 * float(sa.abs().max()))


@pytest.mark.parametrize("M,K,p,generic", [(16384, 512, 0.3, False), (96, 512, 0.0, False), (64, 64, 0.2, True)])
def test_linear_dgrad_ln_backward_with_second_gemm(M, K, p, generic, monkeypatch):
    """mm_linear_dgrad_ln_bwd_gemm2: the rows masked for the out-projection's backward are multiplied by its data-gradient
    image inside the launch - same dx, same masked rows, and do bit-identical to the separate data-gradient launch."""
    hip = _hip()
    if generic:
        monkeypatch.setenv("MM_EPI_GENERIC", "1")
    g = torch.Generator().manual_seed(M + K + 1)
    w = torch.randn(K, 128, 1, generator=g) / math.sqrt(128)                # linear1: 128 -> K, input LN(x1)
    _, wd = _prep_w(hip, w, 128, K)
    wo = torch.randn(128, 128, 1, generator=g) / math.sqrt(128)             # out_proj: 128 -> 128
    _, wdo = _prep_w(hip, wo, 128, 128)
    dy = (torch.randn(M, K, generator=g) * 0.1).cuda().to(torch.bfloat16)
    x = torch.randn(M, 128, generator=g).cuda()
    stat = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    gam = (0.5 + torch.rand(128, generator=g)).cuda()
    dres = (torch.randn(M, 128, generator=g) * 0.1).cuda()

    def run(fused):
        dx = torch.full((M, 128), float("nan"), device="cuda")
        dxb = torch.full((M, 128), float("nan"), device="cuda").to(torch.bfloat16)
        dgb = torch.zeros(32, 2, 128, device="cuda")
        do = torch.full((M, 128), float("nan"), device="cuda").to(torch.bfloat16)
        if fused:
            hip.call("mm_linear_dgrad_ln_bwd_gemm2", dy, wd, M, K, x, stat, gam, dres, dx, dxb, dgb, p, 41, None, wdo, do, 0)
        else:
            hip.call("mm_linear_dgrad_ln_bwd", dy, wd, M, K, x, stat, gam, dres, dx, dxb, dgb, p, 41, None)
            hip.call("mm_conv1d_fwd", dxb, wdo, 1, M, 128, 128, 1, 0, None, None, 0, None, None, 1, None, None, do, None,
                     0.0, 0, None, None, 0)
        return dx, dxb, dgb, do
    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
    assert torch.isfinite(b[3].float()).all() and b[3].float().abs().max() > 1e-3
    assert torch.equal(a[3], b[3])
    ref = (a[1].float().cpu() @ _bf(wo[:, :, 0])).to(torch.bfloat16).float()           # do = dyo @ Wo  (Wo: out x in)
    torch.testing.assert_close(b[3].float().cpu(), ref, rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("R,S,N,p", [(32, 512, 128, 0.3), (3, 40, 48, 0.0)])
def test_bn_act_backward_with_one_dout_row_per_sample(R, S, N, p):
    """mm_bn_act_bwd_reduce_bcast / _apply_bcast = the ordinary passes on the (R, S, N) tensor whose every position holds
    the sample's row times 1 / S (the backward of a mean over positions): same accumulator words, same dy."""
    hip = _hip()
    g = torch.Generator().manual_seed(R * S + N)
    y = (torch.randn(R, S, N, generator=g) * 1.3 + 0.2).cuda()
    out4 = torch.stack([0.5 + torch.rand(N, generator=g), torch.randn(N, generator=g) * 0.2,
                        torch.randn(N, generator=g) * 0.1, 0.8 + 0.4 * torch.rand(N, generator=g)]).cuda().contiguous()
    rows = torch.randn(R, N, generator=g).cuda()
    scale = 1.0 / S
    full = (rows * torch.tensor(scale, dtype=torch.float32, device="cuda")).view(R, 1, N).expand(R, S, N).contiguous()
    sums_a = torch.zeros(32, 2, N, device="cuda")
    dy_a = torch.empty(R, S, N, dtype=torch.bfloat16, device="cuda")
    args = (R, S, N, 1, 1, 1, p, 51, 0.0, 0, None)
    hip.call("mm_bn_act_bwd_reduce", y, out4, None, full, sums_a, *args)
    hip.call("mm_bn_act_bwd_apply", y, out4, None, full, sums_a, dy_a, None, *args, 1, 32)
    sums_b = torch.zeros(32, 2, N, device="cuda")
    dy_b = torch.full((R, S, N), float("nan"), device="cuda").to(torch.bfloat16)
    hip.call("mm_bn_act_bwd_reduce_bcast", y, out4, rows, scale, sums_b, R, S, N, 1, p, 51, None)
    hip.call("mm_bn_act_bwd_apply_bcast", y, out4, rows, scale, sums_b, dy_b, R, S, N, 1, p, 51, None, 1, 32)
    assert torch.equal(sums_a.view(torch.int32), sums_b.view(torch.int32))
    assert torch.equal(dy_a, dy_b) and torch.isfinite(dy_b.float()).all()


def test_pooled_head_backward_row_form_and_row_residual():
    """mm_pooled_head_bwd_rows: the one fp32 row per sample instead of the (B, L, D) token gradients (same dz, same masked
    bf16 tokens); mm_linear_dgrad_ln_bwd_gemm2 fed with that row (dres_rows_per_sample = L) = fed with the expanded tensor."""
    hip = _hip()
    g = torch.Generator().manual_seed(5)
    B, L, D, N = 4, 64, 128, 128
    dout = torch.randn(B, N, generator=g).cuda()
    z = torch.randn(B, N, generator=g).cuda().to(torch.bfloat16)
    W = (torch.randn(N, D, generator=g) / math.sqrt(D)).cuda()
    dz_a = torch.empty(B, N, dtype=torch.bfloat16, device="cuda"); dz_b = torch.empty_like(dz_a)
    dx = torch.empty(B, L, D, device="cuda")
    em_a = torch.empty(B, L, D, dtype=torch.bfloat16, device="cuda"); em_b = torch.full_like(em_a, float("nan"))
    rows = torch.full((B, D), float("nan"), device="cuda")
    hip.call("mm_pooled_head_bwd", dout, z, W, dz_a, dx, em_a, B, L, D, N, 1, 0.3, 61, 0.2, 62, None)
    hip.call("mm_pooled_head_bwd_rows", dout, z, W, dz_b, rows, em_b, B, L, D, N, 1, 0.3, 61, 0.2, 62, None)
    assert torch.equal(dz_a, dz_b) and torch.equal(em_a, em_b)
    assert torch.equal(rows, dx[:, 0, :]) and torch.equal(rows.view(B, 1, D).expand(B, L, D), dx)
    M, K = B * L, 512
    w = torch.randn(K, 128, 1, generator=g) / math.sqrt(128)
    _, wd = _prep_w(hip, w, 128, K)
    _, wdo = _prep_w(hip, torch.randn(128, 128, 1, generator=g) / math.sqrt(128), 128, 128)
    dy = (torch.randn(M, K, generator=g) * 0.1).cuda().to(torch.bfloat16)
    x = torch.randn(M, 128, generator=g).cuda()
    stat = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()], 1).contiguous()
    gam = (0.5 + torch.rand(128, generator=g)).cuda()
    outs = []
    for dres, per in ((dx.view(M, D).contiguous(), 0), (rows, L)):
        o = [torch.full((M, 128), float("nan"), device="cuda"), torch.empty(M, 128, dtype=torch.bfloat16, device="cuda"),
             torch.zeros(32, 2, 128, device="cuda"), torch.empty(M, 128, dtype=torch.bfloat16, device="cuda")]
        hip.call("mm_linear_dgrad_ln_bwd_gemm2", dy, wd, M, K, x, stat, gam, dres, o[0], o[1], o[2], 0.1, 63, None, wdo, o[3], per)
        outs.append(o)
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a, b.view(torch.int32) if b.dtype == torch.float32 else b)
    with pytest.raises(hip.HipLibraryError):                      # rows per sample must divide M
        hip.call("mm_linear_dgrad_ln_bwd_gemm2", dy, wd, M, K, x, stat, gam, rows, outs[0][0], outs[0][1], outs[0][2], 0.1, 63, None,
                 wdo, outs[0][3], 7)


@pytest.mark.parametrize("M,K,p,generic", [(16384, 512, 0.3, False), (96, 128, 0.0, False), (64, 64, 0.2, True)])
def test_linear_forward_with_fused_layernorm_and_next_projection(M, K, p, generic, monkeypatch):
    """mm_linear_fwd_ln_gemm2: the fused LayerNorm's rows go through the next block's QKV projection inside the launch -
    q | k | v bit-identical to the projection as a launch of its own, everything else unchanged."""
    hip = _hip()
    if generic:
        monkeypatch.setenv("MM_EPI_GENERIC", "1")
    g = torch.Generator().manual_seed(M + K + 2)
    w = torch.randn(128, K, 1, generator=g) / math.sqrt(K)
    wf, _ = _prep_w(hip, w, K)
    wq = torch.randn(384, 128, 1, generator=g) / math.sqrt(128)
    wqf, _ = _prep_w(hip, wq, 128)
    bq = torch.randn(384, generator=g).cuda() * 0.1
    x = (torch.randn(M, K, generator=g) * 0.5).cuda().to(torch.bfloat16)
    bias = torch.randn(128, generator=g).cuda() * 0.1
    res = torch.randn(M, 128, generator=g).cuda()
    gam, bet = (0.5 + torch.rand(128, generator=g)).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()

    def run(fused):
        o = torch.full((M, 128), float("nan"), device="cuda")
        h = torch.empty(M, 128, dtype=torch.bfloat16, device="cuda")
        st = torch.empty(M, 2, device="cuda")
        q = torch.full((M, 384), float("nan"), device="cuda").to(torch.bfloat16)
        if fused:
            hip.call("mm_linear_fwd_ln_gemm2", x, wf, M, K, bias, res, o, p, 71, None, gam, bet, 1e-5, h, st, wqf, bq, 384, q)
        else:
            hip.call("mm_linear_fwd_ln", x, wf, M, K, bias, res, o, p, 71, None, gam, bet, 1e-5, h, st)
            hip.call("mm_conv1d_fwd", h, wqf, 1, M, 128, 384, 1, 0, None, bq, 0, None, None, 1, None, None, q, None,
                     0.0, 0, None, None, 0)
        return o, h, st, q
    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert torch.isfinite(b[3].float()).all() and torch.equal(a[3], b[3])
    ref = (a[1].float().cpu() @ _bf(wq[:, :, 0]).t() + bq.cpu()).to(torch.bfloat16).float()
    torch.testing.assert_close(b[3].float().cpu(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M,K,p,generic", [(16384, 128, 0.3, False), (96, 128, 0.0, False), (64, 64, 0.2, True)])
def test_linear_forward_with_fused_layernorm_and_first_ffn_linear(M, K, p, generic, monkeypatch):
    """mm_linear_fwd_ln_gemm2_act: out-projection + norm2, then Linear(128 -> 512) + GELU + dropout on norm2's rows in the
    same launch - activation and pre-activation bit-identical to the Linear as a launch of its own."""
    hip = _hip()
    if generic:
        monkeypatch.setenv("MM_EPI_GENERIC", "1")
    g = torch.Generator().manual_seed(M + K + 3)
    wf, _ = _prep_w(hip, torch.randn(128, K, 1, generator=g) / math.sqrt(K), K)
    w1 = torch.randn(512, 128, 1, generator=g) / math.sqrt(128)
    w1f, _ = _prep_w(hip, w1, 128)
    b1 = torch.randn(512, generator=g).cuda() * 0.1
    x = (torch.randn(M, K, generator=g) * 0.5).cuda().to(torch.bfloat16)
    bias = torch.randn(128, generator=g).cuda() * 0.1
    res = torch.randn(M, 128, generator=g).cuda()
    gam, bet = (0.5 + torch.rand(128, generator=g)).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()

    def run(fused):
        o = torch.full((M, 128), float("nan"), device="cuda")
        h = torch.empty(M, 128, dtype=torch.bfloat16, device="cuda")
        st = torch.empty(M, 2, device="cuda")
        a1 = torch.full((M, 512), float("nan"), device="cuda").to(torch.bfloat16)
        z1 = torch.full((M, 512), float("nan"), device="cuda").to(torch.bfloat16)
        if fused:
            hip.call("mm_linear_fwd_ln_gemm2_act", x, wf, M, K, bias, res, o, p, 81, None, gam, bet, 1e-5, h, st, w1f, b1, 512,
                     a1, z1, 1, p, 82)
        else:
            hip.call("mm_linear_fwd_ln", x, wf, M, K, bias, res, o, p, 81, None, gam, bet, 1e-5, h, st)
            hip.call("mm_conv1d_fwd", h, w1f, 1, M, 128, 512, 1, 0, None, b1, 1, None, None, 1, None, None, a1, z1,
                     p, 82, None, None, 0)
        return o, h, st, a1, z1
    a, b = run(False), run(True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert torch.isfinite(b[3].float()).all() and torch.isfinite(b[4].float()).all()
    zref = (a[1].float().cpu() @ _bf(w1[:, :, 0]).t() + b1.cpu())
    torch.testing.assert_close(b[4].float().cpu(), zref.to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2)


def _vol_cl(x):           # (B,C,D,H,W) -> channels-last bf16 on GPU
    return x.permute(0, 2, 3, 4, 1).contiguous().cuda().to(torch.bfloat16)


@pytest.mark.parametrize("B,Cin,Cout,D,H,W", [(2, 32, 64, 8, 8, 8), (1, 64, 128, 4, 8, 8), (2, 16, 32, 8, 16, 8),
                                              (1, 32, 64, 6, 8, 12)])
def test_conv3d_fwd_wgrad_dgrad(B, Cin, Cout, D, H, W):
    hip = _hip()
    g = torch.Generator().manual_seed(Cin + Cout + D)
    x = _bf(torch.randn(B, Cin, D, H, W, generator=g)).requires_grad_(True)
    w = _bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin)).requires_grad_(True)
    bias = torch.randn(Cout, generator=g).requires_grad_(True)
    dy = _bf(torch.randn(B, Cout, D, H, W, generator=g))
    y = F.conv3d(x, w, bias, padding=1)
    y.backward(dy)
    wf, wd = _prep_w(hip, w.detach().reshape(Cout, Cin, 27), Cin, Cout)
    xg = _vol_cl(x.detach())
    out = torch.empty(B, D, H, W, Cout, device="cuda")
    stats = torch.zeros(32, 2, Cout, device="cuda")
    hip.call("mm_conv3d_fwd", xg, wf, B, D, H, W, Cin, Cout, bias.detach().cuda(), stats, out, None)
    stats = _stat(stats)
    want = y.detach().permute(0, 2, 3, 4, 1)
    torch.testing.assert_close(out.cpu(), want, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(stats[0].cpu(), want.sum(dim=(0, 1, 2, 3)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(stats[1].cpu(), (want * want).sum(dim=(0, 1, 2, 3)), rtol=1e-3, atol=1e-2)
    dyg = _vol_cl(dy)
    # plain stores into NaN-filled per-chunk slots; first with the strides of the PyTorch layout
    import ctypes
    n = ctypes.c_int(0)
    hip.call("mm_conv3d_wgrad_slots", B, D, H, W, Cin, Cout, ctypes.addressof(n))
    dw = torch.full((n.value, Cout, Cin, 27), float("nan"), device="cuda")
    db = torch.zeros(32, Cout, device="cuda")
    hip.call("mm_conv3d_wgrad", dyg, xg, dw, db, B, D, H, W, Cin, Cout, Cin, Cin * 27, 27, 1, n.value, Cout * Cin * 27, 1)
    dw, db = dw.sum(0), _grad(db)
    torch.testing.assert_close(dw.cpu().view_as(w), w.grad, rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(db.cpu(), bias.grad, rtol=2e-3, atol=2e-2)
    with pytest.raises(hip.HipLibraryError, match="slot_mode"):          # the fp32-atomics mode is gone
        hip.call("mm_conv3d_wgrad", dyg, xg, dw, db, B, D, H, W, Cin, Cout, Cin, Cin * 27, 27, 1, 3, Cout * Cin * 27, 0)
    # channel-contiguous slots, summed by the scatter
    ws = torch.full((n.value, Cout, 27, Cin), float("nan"), device="cuda")
    hip.call("mm_conv3d_wgrad", dyg, xg, ws, None, B, D, H, W, Cin, Cout, Cin, 27 * Cin, 1, Cin, n.value, Cout * 27 * Cin, 1)
    dw2 = torch.zeros(Cout, Cin, 27, device="cuda")
    hip.call("mm_wgrad_scatter", ws, dw2, Cout, Cin, 27, Cin, n.value)
    torch.testing.assert_close(dw2.cpu().view_as(w), w.grad, rtol=2e-3, atol=2e-2)
    dx = torch.empty(B, D, H, W, Cin, device="cuda")
    hip.call("mm_conv3d_fwd", dyg, wd, B, D, H, W, Cout, Cin, None, None, dx, None)
    torch.testing.assert_close(dx.cpu(), x.grad.permute(0, 2, 3, 4, 1), rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("scale,expect", [(30.0, "exact"), (3e3, "flagged"), (float("nan"), "flagged"), (float("inf"), "flagged")])
def test_accumulator_overflow_is_reported_not_wrapped(scale, expect):
    """csrc/common.h: the 64-bit fixed-point accumulators either hold the exact sum or read as NaN (a contribution
    out of range / non-finite poisons its replica; replica magnitudes reaching 2^61 flag the sum) - never a wrapped
    integer.  Through a real producer / consumer pair: the weight-resident conv3d's BatchNorm sums -> mm_bn_finalize.
    scale 30: sum of squares ~1.5e7, in range -> statistics to 1e-3; scale 3e3: ~1.5e11 > 8.6e9 -> NaN mean / rstd
    and untouched-by-garbage running statistics are not required, only non-finite ones; NaN / Inf inputs -> NaN."""
    from multimodal_eeg_fmri_amd.ops import ACC_STAT, acc_decode
    hip = _hip()
    B, Cin, Cout, D, H, W = 4, 32, 64, 16, 16, 16
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(B, Cin, D, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    if math.isfinite(scale):
        x = _bf(x * scale)
    else:
        x[1, 3, 4, 5, 6] = scale
    wf, _ = _prep_w(hip, w.reshape(Cout, Cin, 27), Cin)
    out = torch.empty(B, D, H, W, Cout, dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(32, 2, Cout, device="cuda")
    hip.call("mm_conv3d_fwd", _vol_cl(x), wf, B, D, H, W, Cin, Cout, None, stats, None, out)
    dec = acc_decode(stats, ACC_STAT).cpu()
    gam, bet = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    out4 = torch.empty(4, Cout, device="cuda")
    hip.call("mm_bn_finalize", stats, gam, bet, rm, rv, None, out4, Cout, float(B * D * H * W), 0.1, 1e-5, 0, None)
    if expect == "exact":
        want = F.conv3d(x, w, None, padding=1).permute(0, 2, 3, 4, 1).double()
        torch.testing.assert_close(dec[0], want.sum(dim=(0, 1, 2, 3)), rtol=1e-3, atol=1.0)
        torch.testing.assert_close(dec[1], (want * want).sum(dim=(0, 1, 2, 3)), rtol=1e-3, atol=1.0)
        assert torch.isfinite(out4).all().item()
        torch.testing.assert_close(out4[2].cpu().double(), want.mean(dim=(0, 1, 2, 3)), rtol=1e-3, atol=1e-3)
    else:
        assert torch.isnan(dec[1]).any().item()                       # the sum of squares is out of range / poisoned ...
        bad = torch.isnan(dec).any(dim=0)
        assert torch.isnan(out4[:2].cpu())[:, bad].all().item()       # ... and the consumer says so: scale / shift are NaN
        good = ~bad
        assert torch.isfinite(out4[:, good]).all().item()             # channels whose sums are in range stay exact


def test_accumulator_single_writer_and_lane_sums_report_overflow():
    """the other consumer forms: mm_colstats (one writer, acc_encode) -> mm_bn_finalize, and mm_acc_reduce
    (one replica per lane, acc_sum_lanes16) on a hand-made workspace holding a poisoned replica, a wrapped-range
    sum and an in-range sum."""
    from multimodal_eeg_fmri_amd.ops import ACC_GRAD
    hip = _hip()
    N = 16
    x = torch.randn(8, N)
    x[:, 3] *= 1e6
    x[2, 5] = float("nan")
    stats = torch.zeros(32, 2, N, device="cuda")
    hip.call("mm_colstats", x.cuda(), stats, 8, N)
    out4 = torch.empty(4, N, device="cuda")
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    hip.call("mm_bn_finalize", stats, torch.ones(N, device="cuda"), torch.zeros(N, device="cuda"), rm, rv, None, out4, N, 8.0, 0.1,
             1e-5, 0, None)
    o = out4.cpu()
    assert torch.isnan(o[0, 3]).item() and torch.isnan(o[0, 5]).item()
    keep = [i for i in range(N) if i not in (3, 5)]
    assert torch.isfinite(o[:, keep]).all().item()
    torch.testing.assert_close(o[2, keep], x[:, keep].mean(0), rtol=1e-5, atol=1e-6)
    ws = torch.zeros(16, 4, dtype=torch.int64)
    ws[:, 0] = int(1.5 * 2 ** ACC_GRAD)                      # 16 x 1.5 = 24
    ws[5, 1] = 1 << 62                                        # poisoned replica
    ws[:, 2] = 1 << 58                                        # 16 x 2^58 = 2^62: over the flagged range
    ws[:, 3] = -(1 << 57) + 12345                             # 16 x -2^57 = -2^61: flagged (boundary)
    dst = torch.zeros(4, device="cuda")
    hip.call("mm_acc_reduce", ws.view(torch.float32).view(32, 4).cuda(), dst, 4, 4)
    d = dst.cpu()
    assert d[0].item() == 24.0 and torch.isnan(d[1:]).all().item(), d


@pytest.mark.parametrize("B,Cin,Cout,D,H,W", [(32, 32, 64, 16, 16, 16), (32, 64, 128, 8, 8, 8), (8, 32, 64, 32, 32, 24),
                                              (3, 32, 64, 14, 16, 20)])
def test_conv3d_wgrad_at_training_shapes(B, Cin, Cout, D, H, W):
    """csrc/conv3d_wgrad.hip at the shapes the training step runs it at - layer 2 and layer 3 of the C2 step (41 / 25
    slot counts, the 4-way K split, the LDS-DMA ring at full depth), layer 2 at BASELINE config #4 volumes, one ragged
    volume - against F.conv3d's weight gradient on the same bf16-rounded operands: rel-L2 <= 2e-3 and element-wise 2e-3
    of the tensor's scale; per-chunk slots are NaN-filled before the launch (every element of every slot is written)."""
    import ctypes
    hip = _hip()
    g = torch.Generator().manual_seed(B + Cin + D + W)
    x = _bf(torch.randn(B, Cin, D, H, W, generator=g))
    dy = _bf(torch.randn(B, Cout, D, H, W, generator=g))
    want = torch.nn.grad.conv3d_weight(x, (Cout, Cin, 3, 3, 3), dy, padding=1).reshape(Cout, Cin, 27)
    want_db = dy.sum(dim=(0, 2, 3, 4))
    xg, dyg = _vol_cl(x), _vol_cl(dy)
    n = ctypes.c_int(0)
    hip.call("mm_conv3d_wgrad_slots", B, D, H, W, Cin, Cout, ctypes.addressof(n))
    ws = torch.full((n.value, Cout, 27, Cin), float("nan"), device="cuda")
    db = torch.zeros(32, Cout, device="cuda")
    hip.call("mm_conv3d_wgrad", dyg, xg, ws, db, B, D, H, W, Cin, Cout, Cin, 27 * Cin, 1, Cin, n.value, Cout * 27 * Cin, 1)
    assert torch.isfinite(ws).all().item()
    dw = torch.zeros(Cout, Cin, 27, device="cuda")
    hip.call("mm_wgrad_scatter", ws, dw, Cout, Cin, 27, Cin, n.value)
    got = dw.cpu()
    rel = ((got - want).norm() / want.norm()).item()
    assert rel <= 2e-3, rel
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-3 * scale
    torch.testing.assert_close(_grad(db).cpu(), want_db, rtol=2e-3, atol=2e-3 * want_db.abs().max().item())
    ws2 = torch.full_like(ws, float("nan"))                     # bit-reproducible: slots are plain stores in a fixed order
    hip.call("mm_conv3d_wgrad", dyg, xg, ws2, None, B, D, H, W, Cin, Cout, Cin, 27 * Cin, 1, Cin, n.value, Cout * 27 * Cin, 1)
    assert torch.equal(ws, ws2)


@pytest.mark.parametrize("B,D,H,W", [(4, 16, 16, 16), (32, 16, 16, 16), (8, 32, 32, 24), (24, 14, 16, 20), (2, 16, 24, 20),
                                     (32, 8, 8, 8), (64, 8, 8, 8), (8, 4, 32, 32), (64, 4, 8, 8), (70, 3, 7, 5)])
def test_conv3d_weight_resident_kernel_vs_torch(B, D, H, W):
    """csrc/conv3d_wres.hip (Cin 32 -> Cout 64, bf16 out + BatchNorm sums; layer 2 of the voxel encoder,
    bench.py's roofline kernel) against F.conv3d on bf16-rounded operands: one tile per workgroup (64 tiles),
    the C2 shape (512 tiles: two per workgroup, the second one's K loop carries the first one's stores), the
    config-#4 shape (768 tiles), ragged tiles (14 = 3.5 tiles deep, 20 = 2.5 wide; stored by the predicated
    path) mixed with interior ones, the XCD-aware tile lists, and volumes that are ONE tile wide / high / deep
    (8^3 = layer 2 of a 16^3 volume at B >= 32; 4 x 32 x 32; a single ragged tile per sample: the tile-index
    division by 1 has no 32-bit reciprocal).  bf16 output: 1e-2; statistics come from the
    fp32 accumulators: 1e-3."""
    hip = _hip()
    Cin, Cout = 32, 64
    assert B * ((D + 3) // 4) * ((H + 7) // 8) * ((W + 7) // 8) >= 64       # else the generic kernel runs
    g = torch.Generator().manual_seed(D + H + W)
    x = _bf(torch.randn(B, Cin, D, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    bias = torch.randn(Cout, generator=g)
    want = F.conv3d(x, w, bias, padding=1).permute(0, 2, 3, 4, 1)
    wf, _ = _prep_w(hip, w.reshape(Cout, Cin, 27), Cin)
    xg = _vol_cl(x)
    out = torch.full((B, D, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(32, 2, Cout, device="cuda")
    hip.call("mm_conv3d_fwd", xg, wf, B, D, H, W, Cin, Cout, bias.cuda(), stats, None, out)
    torch.testing.assert_close(out.float().cpu(), want, rtol=1e-2, atol=1e-2)
    st = _stat(stats).cpu()
    torch.testing.assert_close(st[0], want.sum(dim=(0, 1, 2, 3)), rtol=1e-3, atol=5e-2)
    torch.testing.assert_close(st[1], (want * want).sum(dim=(0, 1, 2, 3)), rtol=1e-3, atol=5e-2)
    out2 = torch.empty_like(out)                                        # no bias, no statistics (eval form)
    hip.call("mm_conv3d_fwd", xg, wf, B, D, H, W, Cin, Cout, None, None, None, out2)
    torch.testing.assert_close(out2.float().cpu(), want - bias, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("B,Cin,Cout,D,H,W", [(32, 64, 128, 8, 8, 8), (2, 128, 64, 8, 8, 8), (3, 64, 32, 5, 16, 16),
                                              (2, 64, 128, 3, 12, 20), (1, 128, 64, 2, 20, 12), (2, 64, 32, 4, 10, 9)])
def test_conv3d_weight_streaming_kernel_vs_torch(B, Cin, Cout, D, H, W):
    """csrc/conv3d_stream.hip (layer 3 forward 64 -> 128, its data gradient 128 -> 64, layer 2's data gradient
    64 -> 32: one 1 x 8 x 8 face per workgroup, weights streamed through an LDS ring by LDS-DMA, K-groups summed
    through LDS) against F.conv3d on bf16-rounded operands: the C2 shapes, ragged faces (12, 20, 10, 9 wide),
    grids that are / are not a multiple of 8 (XCD-aware tile order on / off), fp32 and bf16 outputs, BatchNorm sums."""
    hip = _hip()
    g = torch.Generator().manual_seed(Cin + Cout + D + H)
    x = _bf(torch.randn(B, Cin, D, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, 3, 3, 3, generator=g) / math.sqrt(27 * Cin))
    bias = torch.randn(Cout, generator=g)
    want = F.conv3d(x, w, bias, padding=1).permute(0, 2, 3, 4, 1)
    wf, _ = _prep_w(hip, w.reshape(Cout, Cin, 27), Cin)
    xg = _vol_cl(x)
    out = torch.full((B, D, H, W, Cout), float("nan"), device="cuda")
    outb = torch.full((B, D, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(32, 2, Cout, device="cuda")
    hip.call("mm_conv3d_fwd", xg, wf, B, D, H, W, Cin, Cout, bias.cuda(), stats, out, outb)
    torch.testing.assert_close(out.cpu(), want, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(outb.float().cpu(), want, rtol=1e-2, atol=1e-2)
    st = _stat(stats).cpu()
    torch.testing.assert_close(st[0], want.sum(dim=(0, 1, 2, 3)), rtol=1e-3, atol=5e-2)
    torch.testing.assert_close(st[1], (want * want).sum(dim=(0, 1, 2, 3)), rtol=1e-3, atol=5e-2)
    out2 = torch.full_like(outb, float("nan"))                         # the data-gradient form: no bias, no statistics, bf16 out
    hip.call("mm_conv3d_fwd", xg, wf, B, D, H, W, Cin, Cout, None, None, None, out2)
    torch.testing.assert_close(out2.float().cpu(), want - bias, rtol=1e-2, atol=1e-2)


def test_pool3d_bn_act_train_fwd_bwd():
    hip = _hip()
    g = torch.Generator().manual_seed(9)
    B, D, H, W, N = 2, 4, 8, 4, 32
    y = _bf(torch.randn(B, D, H, W, N, generator=g) * 1.5 + 0.3)      # the conv's pre-BN output is kept in bf16
    gam = 0.5 + torch.rand(N, generator=g)
    bet = torch.randn(N, generator=g) * 0.2
    yr, gr, br = y.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    z = F.batch_norm(yr.permute(0, 4, 1, 2, 3), None, None, gr, br, training=True, eps=1e-5)
    a = F.max_pool3d(F.gelu(z), 2).permute(0, 2, 3, 4, 1)
    dout = _bf(torch.randn(a.shape, generator=g))
    a.backward(dout)
    yg = y.cuda().to(torch.bfloat16)
    flat = y.reshape(-1, N)
    from multimodal_eeg_fmri_amd.ops import ACC_STAT, acc_encode
    stats = acc_encode(torch.stack([flat.sum(0), (flat * flat).sum(0)]), ACC_STAT).cuda()
    out4 = torch.empty(4, N, device="cuda")
    hip.call("mm_bn_finalize", stats, gam.cuda(), bet.cuda(), torch.zeros(N, device="cuda"), torch.ones(N, device="cuda"),
             None, out4, N, float(flat.shape[0]), 0.1, 1e-5, 0, None)
    ob = torch.empty(B, D // 2, H // 2, W // 2, N, dtype=torch.bfloat16, device="cuda")
    ysel = torch.empty(ob.shape, dtype=torch.bfloat16, device="cuda")
    arg = torch.empty(ob.shape, dtype=torch.uint8, device="cuda")
    hip.call("mm_pool3d_bn_act_fwd", yg, out4, ob, ysel, arg, B, D, H, W, N, 1, 0.0, 0, None)
    torch.testing.assert_close(ob.float().cpu(), a.detach(), rtol=1e-2, atol=1e-2)
    # the saved winners: window index (4 d + 2 h + w) and pre-BN value of torch's own argmax
    _, idx = F.max_pool3d(F.gelu(z.detach()), 2, return_indices=True)          # flat index into D*H*W
    idx = idx.permute(0, 2, 3, 4, 1)
    d_, h_, w_ = idx // (H * W), (idx // W) % H, idx % W
    torch.testing.assert_close(arg.cpu().long(), (d_ % 2) * 4 + (h_ % 2) * 2 + (w_ % 2))
    want = torch.gather(y.reshape(B, D * H * W, N), 1, idx.reshape(B, -1, N)).reshape(ob.shape)
    torch.testing.assert_close(ysel.float().cpu(), want)
    ob2 = torch.empty_like(ob)                                                  # eval form: no winners kept
    hip.call("mm_pool3d_bn_act_fwd", yg, out4, ob2, None, None, B, D, H, W, N, 1, 0.0, 0, None)
    assert torch.equal(ob2, ob)
    sums = torch.zeros(32, 2, N, device="cuda")
    dg = dout.cuda().to(torch.bfloat16)
    hip.call("mm_pool3d_bn_act_bwd_reduce", ysel, out4, dg, sums, B, D, H, W, N, 1, 0.0, 0, None)
    torch.testing.assert_close(_grad(sums)[0].cpu(), br.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(_grad(sums)[1].cpu(), gr.grad, rtol=1e-3, atol=1e-3)
    dy = torch.empty(B, D, H, W, N, dtype=torch.bfloat16, device="cuda")
    sc = torch.zeros(2 * N, device="cuda")
    hip.call("mm_acc_reduce", sums, sc, 2 * N, 2 * N)
    torch.testing.assert_close(sc.cpu(), _grad(sums).flatten().cpu(), rtol=1e-6, atol=1e-7)
    sc2 = torch.ones(N, device="cuda")                      # offset e of a workspace = byte offset 8 e; dst is ADDED to
    hip.call("mm_acc_reduce", sums.data_ptr() + 8 * N, sc2, N, 2 * N)
    torch.testing.assert_close(sc2.cpu(), 1 + _grad(sums)[1].cpu(), rtol=1e-6, atol=1e-6)
    fl = torch.arange(3 * 5, dtype=torch.float32, device="cuda").view(3, 5)      # the fp32 form: replicas summed in order
    d3 = torch.zeros(5, device="cuda")
    hip.call("mm_reduce_replicas", fl, d3, 5, 3, 5)
    assert torch.equal(d3, fl.sum(0))
    hip.call("mm_pool3d_bn_act_bwd_apply", yg, arg, out4, dg, sc, dy, B, D, H, W, N, 1, 0.0, 0, None, 1, 1)
    torch.testing.assert_close(dy.float().cpu(), yr.grad, rtol=2e-2, atol=2e-3)
    dy2 = torch.empty_like(dy)                      # same, the kernel summing the workspace's replicas itself
    hip.call("mm_pool3d_bn_act_bwd_apply", yg, arg, out4, dg, sums, dy2, B, D, H, W, N, 1, 0.0, 0, None, 1, 32)
    assert torch.equal(dy2, dy)



@pytest.mark.parametrize("L", [64, 77, 256])      # 77: ragged tiles, odd row length (pairs do not straddle rows); 256: the full-tile specialisation (L % 256 == 0) of all three kernels
def test_attention_dropout_is_consistent_between_fwd_and_bwd(L):
    """attention-probability dropout: forward equals softmax(S) * mask / keep @ V for the
    kernel's own hash mask, and backward differentiates exactly that function
    (checked with a finite-difference directional derivative in fp64 on the CPU)."""
    hip = _hip()
    B, H, p, seed = 1, 2, 0.3, 1234
    E = H * 32
    g = torch.Generator().manual_seed(3)
    qkv = _bf(torch.randn(B, L, 3 * E, generator=g) * 0.5)
    qg = qkv.cuda().to(torch.bfloat16)
    out = torch.empty(B, L, E, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, L, device="cuda")
    hip.call("mm_attn_fwd", qg, out, lse, B, L, H, 32, 1 / math.sqrt(32), p, seed, None, None, 0)
    out0 = torch.empty_like(out)
    hip.call("mm_attn_fwd", qg, out0, lse, B, L, H, 32, 1 / math.sqrt(32), 0.0, 0, None, None, 0)
    # recover the mask from V = identity-like probe: compare row sums of kept probabilities
    # host replica of the attention kernels' block hash (attention.hip: attn_block_hash / attn_keep2)
    from oracle.dropout_replica import attn_keep_scale
    m = attn_keep_scale(seed, B * H, L, p).view(B, H, L, L).double()
    q, k, v = (t.view(B, L, H, 32).transpose(1, 2).double() for t in qkv.split(E, dim=2))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(32)
    o_ref = ((torch.softmax(s, -1) * m) @ v).transpose(1, 2).reshape(B, L, E)
    torch.testing.assert_close(out.float().cpu(), o_ref.detach().float(), rtol=3e-2, atol=3e-2)
    assert (out.float() - out0.float()).abs().max().item() > 1e-2          # dropout really happened
    do = _bf(torch.randn(B, L, E, generator=g))
    o_ref.backward(do.double())
    dqkv = torch.empty_like(qg)
    delta = torch.empty(B, H, L, device="cuda")
    hip.call("mm_attn_bwd", qg, out, do.cuda().to(torch.bfloat16), lse, dqkv, delta, B, L, H, 32, 1 / math.sqrt(32), p, seed, None, None, 0)
    dq_got = dqkv.float().cpu()[:, :, :E].view(B, L, H, 32).transpose(1, 2)
    assert ((dq_got - q.grad.float()).norm() / q.grad.float().norm()).item() < 3e-2
    # the dkv kernel's registers run along the queries: it reads the other two bytes of each 2 x 2 block hash
    dk_got = dqkv.float().cpu()[:, :, E:2 * E].view(B, L, H, 32).transpose(1, 2)
    dv_got = dqkv.float().cpu()[:, :, 2 * E:].view(B, L, H, 32).transpose(1, 2)
    assert ((dk_got - k.grad.float()).norm() / k.grad.float().norm()).item() < 3e-2
    assert ((dv_got - v.grad.float()).norm() / v.grad.float().norm()).item() < 3e-2


def test_batched_launches_equal_their_single_forms():
    """mm_prep_many_zero == one mm_prep_conv_weight per tensor (workgroups dealt out by tensor size; 70 tensors also
    exercise the split into tables of 64) + a cleared range; mm_flush_many == mm_scatter_many + mm_reduce_many."""
    import ctypes
    import struct
    hip = _hip()
    g = torch.Generator().manual_seed(8)
    shapes = [(128, 64, 27), (64, 32, 27), (512, 128, 1), (128, 512, 1), (384, 128, 1), (64, 7, 7), (32, 1, 27)] + [(24, 40, 3)] * 63
    keep, raw, want = [], [], []
    for cout, cin, k in shapes:
        cinp, coutp = _cpad(cin), _cpad(cout)
        w = torch.randn(cout, cin, k, generator=g).cuda()
        wf_a = torch.empty(cout, k, cinp, dtype=torch.bfloat16, device="cuda")
        wd_a = torch.empty(cinp, k, coutp, dtype=torch.bfloat16, device="cuda")
        hip.call("mm_prep_conv_weight", w, wf_a, wd_a, cout, cin, k, cinp, coutp)
        wf_b, wd_b = torch.full_like(wf_a, float("nan")), torch.full_like(wd_a, float("nan"))
        raw.append(struct.pack("<QQQiiiiii", w.data_ptr(), wf_b.data_ptr(), wd_b.data_ptr(), cout, cin, k, cinp, coutp, 0))
        keep.append(w)
        want.append((wf_a, wd_a, wf_b, wd_b))
    arena = torch.full((4096 + 8,), 7.0, device="cuda")
    buf = b"".join(raw)
    host = ctypes.create_string_buffer(buf, len(buf))
    hip.call("mm_prep_many_zero", ctypes.addressof(host), len(shapes), arena, 4096)
    torch.cuda.synchronize()
    for wf_a, wd_a, wf_b, wd_b in want:
        assert torch.equal(wf_a.view(torch.int16), wf_b.view(torch.int16)) and torch.equal(wd_a.view(torch.int16), wd_b.view(torch.int16))
    assert (arena[:4096] == 0).all() and (arena[4096:] == 7.0).all()
    with pytest.raises(hip.HipLibraryError):
        hip.call("mm_prep_many_zero", ctypes.addressof(host), len(shapes), arena, 4095)      # not a multiple of 4 floats
    # gradient flush: six slot workspaces + five reductions (accumulator workspaces and one compact fp32 vector)
    from multimodal_eeg_fmri_amd.ops import ACC_GRAD, acc_encode
    sdesc, rdesc, ref = [], [], []
    # (the 300- and 1 000-channel ones take mm_flush_many's LDS-transposing form for wide layers; mm_wgrad_scatter is the plain form)
    for cout, cin, k, slots in [(64, 48, 5, 7), (128, 128, 1, 3), (32, 20, 3, 1), (24, 300, 7, 3), (40, 1000, 5, 9), (190, 1600, 7, 2)]:      # (the last: > 2 M weights, dealt out as three descriptors)
        cinp = _cpad(cin)
        ws = torch.randn(slots, cout, k, cinp, generator=g).cuda()
        dw_a, dw_b = torch.ones(cout, cin, k, device="cuda"), torch.ones(cout, cin, k, device="cuda")
        hip.call("mm_wgrad_scatter", ws, dw_a, cout, cin, k, cinp, slots)
        sdesc.append(struct.pack("<QQiiiiii", ws.data_ptr(), dw_b.data_ptr(), cout, cin, k, cinp, slots, 0))
        keep.append(ws)
        ref.append((dw_a, dw_b))
    for K in (128, 64, 1, 300):
        acc = acc_encode(torch.randn(2, K, generator=g) * 1e-2, ACC_GRAD).cuda()
        d_a, d_b = torch.ones(K, device="cuda"), torch.ones(K, device="cuda")
        hip.call("mm_acc_reduce", acc.data_ptr() + 8 * K, d_a, K, 2 * K)
        rdesc.append(struct.pack("<QQqqq", acc.data_ptr() + 8 * K, d_b.data_ptr(), K, 16, 2 * K))
        keep.append(acc)
        ref.append((d_a, d_b))
    vec = torch.randn(77, generator=g).cuda()
    d_a, d_b = torch.ones(77, device="cuda"), torch.ones(77, device="cuda")
    d_a += vec
    rdesc.append(struct.pack("<QQqqq", vec.data_ptr(), d_b.data_ptr(), 77, 1, 77))
    ref.append((d_a, d_b))
    sb, rb = b"".join(sdesc), b"".join(rdesc)
    sh, rh = ctypes.create_string_buffer(sb, len(sb)), ctypes.create_string_buffer(rb, len(rb))
    hip.call("mm_flush_many", ctypes.addressof(sh), len(sdesc), ctypes.addressof(rh), len(rdesc))
    torch.cuda.synchronize()
    for a_, b_ in ref:
        assert torch.equal(a_, b_)
    with pytest.raises(hip.HipLibraryError):
        hip.call("mm_flush_many", None, 0, None, 0)
    # descriptors that take a BLOCK of a wider workspace (the k = 3 / 5 / 7 branches of EnhancedPowerEncoder's merged k = 7
    # convolution: 64 output channels each, centred tap windows), narrow (plain form) and wide (LDS-transposing form)
    from multimodal_eeg_fmri_amd.autograd import scatter_desc
    for cin in (40, 300):
        cinp, slots = _cpad(cin), 2
        ws = torch.randn(slots, 192, 7, cinp, generator=g).cuda()
        outs, raw = [], []
        for i, kk in enumerate((3, 5, 7)):
            dw = torch.ones(64, cin, kk, device="cuda")
            raw.append(struct.pack("<QQiiiiii", *scatter_desc(ws, dw, 64, cin, kk, cinp, slots, window=(64 * i, (7 - kk) // 2, 7, 192))))
            outs.append((dw, i, kk))
        sb = b"".join(raw)
        sh = ctypes.create_string_buffer(sb, len(sb))
        hip.call("mm_flush_many", ctypes.addressof(sh), 3, None, 0)
        tot = ws[0] + ws[1]
        for dw, i, kk in outs:
            lo = (7 - kk) // 2
            want_w = 1.0 + tot[64 * i:64 * i + 64, lo:lo + kk, :cin].permute(0, 2, 1)
            assert torch.equal(dw, want_w), (cin, kk)
    bad = struct.pack("<QQiiiiii", ws.data_ptr(), outs[0][0].data_ptr(), 64, 300, 5 | (3 << 8) | (7 << 16), 304, 2, 192)   # taps 3..7 of 7
    bh = ctypes.create_string_buffer(bad, len(bad))
    with pytest.raises(hip.HipLibraryError):
        hip.call("mm_flush_many", ctypes.addressof(bh), 1, None, 0)


@pytest.mark.parametrize("B,C,T", [(3, 64, 1024), (2, 20, 77), (1, 16, 32)])
def test_stage_inputs_equals_pack_plus_copies(B, C, T):
    """mm_stage_inputs (one launch in front of a captured step) == mm_pack_nct_bf16 of the EEG batch + an fp32 copy of
    it + an fp32 copy of the fMRI batch, bit for bit; ragged channel / time counts hit the partial 32 x 32 tiles."""
    hip = _hip()
    g = torch.Generator().manual_seed(B * 100 + C + T)
    eeg = torch.randn(B, C, T, generator=g).cuda()
    fmri = torch.randn(B, 1, 8, 8, 12, generator=g).cuda()
    cp = _cpad(C)
    want = torch.empty(B, T, cp, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pack_nct_bf16", eeg, want, B, C, T, cp)
    xb = torch.full((B, T, cp), float("nan"), dtype=torch.bfloat16, device="cuda")
    e2, f2 = torch.full_like(eeg, float("nan")), torch.full_like(fmri, float("nan"))
    hip.call("mm_stage_inputs", eeg, xb, e2, B, C, T, cp, f2, fmri, fmri.numel())
    assert torch.equal(xb.view(torch.int16), want.view(torch.int16)) and torch.equal(e2, eeg) and torch.equal(f2, fmri)
    xb2 = torch.full_like(xb, float("nan"))
    hip.call("mm_stage_inputs", eeg, xb2, None, B, C, T, cp, f2, fmri, fmri.numel())      # without the fp32 EEG copy
    assert torch.equal(xb2.view(torch.int16), want.view(torch.int16))
    with pytest.raises(hip.HipLibraryError):
        hip.call("mm_stage_inputs", eeg, xb2, None, B, C, T, cp, f2, fmri, fmri.numel() - 2)


@pytest.mark.parametrize("B,C,T,nfft,hop", [(2, 3, 256, 64, 32), (1, 2, 1024, 128, 32), (2, 2, 200, 32, 16)])
def test_stft_power_backward_vs_torch_autograd(B, C, T, nfft, hop):
    """mm_stft_power_bwd (gather form, no atomics: every sample sums the frames - and the reflect-padding images - that read
    it, in a fixed order) against autograd through torch.stft(center=True, reflect, periodic Hann) ** 2, and
    mm_sample_zscore_bwd against autograd through the population-std z-score; fp32: 1e-3 of the tensor's scale; two runs
    bit-identical."""
    hip = _hip()
    g = torch.Generator().manual_seed(T + nfft)
    x = torch.randn(B, C, T, generator=g)
    F_ = nfft // 2 + 1
    frames = T // hop + 1
    cp = (C * F_ + 15) // 16 * 16 + 16                       # a channel window inside a wider (padded) tensor
    off = 5
    gP = torch.randn(B, frames, cp, generator=g)
    xr = x.clone().requires_grad_(True)
    z = torch.stft(xr.reshape(B * C, T), nfft, hop_length=hop, window=torch.hann_window(nfft, periodic=True), center=True,
                   pad_mode="reflect", return_complex=True)
    pw = (z.real ** 2 + z.imag ** 2).reshape(B, C * F_, frames).transpose(1, 2)          # (B, frames, C * F)
    (pw * gP[:, :, off:off + C * F_]).sum().backward()
    dx = torch.zeros(B, C, T, device="cuda")
    hip.call("mm_stft_power_bwd", x.cuda(), gP.cuda(), dx, B, C, T, nfft, hop, off, cp)
    scale = xr.grad.abs().max().item()
    assert (dx.cpu() - xr.grad).abs().max().item() <= 1e-3 * scale
    dx2 = torch.zeros_like(dx)
    hip.call("mm_stft_power_bwd", x.cuda(), gP.cuda(), dx2, B, C, T, nfft, hop, off, cp)
    assert torch.equal(dx, dx2)
    # z-score backward: rows x ch_valid valid elements inside ch_total
    rows, chv, cht = 9, 20, 32
    s = torch.randn(B, rows, cht, generator=g) * 3 + 1
    gz = _bf(torch.randn(B, rows, cht, generator=g))
    sr = s.clone().requires_grad_(True)
    v = sr[:, :, :chv]
    flat = v.reshape(B, -1)
    y = (v - flat.mean(1).view(B, 1, 1)) / (flat.std(1, unbiased=False).view(B, 1, 1) + 1e-8)
    (y * gz[:, :, :chv]).sum().backward()
    ds = torch.full((B, rows, cht), float("nan"), device="cuda")
    hip.call("mm_sample_zscore_bwd", s.cuda(), gz.cuda().to(torch.bfloat16), ds, B, rows, chv, cht, 1e-8)
    torch.testing.assert_close(ds.cpu(), sr.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1000, 2_600_000])
def test_adamw_clip_kernel_vs_torch_and_its_bookkeeping(n):
    """mm_sumsq + mm_adamw_clip (run_training_lite.py:487-488: clip_grad_norm_ + AdamW.step) against torch.optim.AdamW on
    the same flat tensor for three steps (1e-6), and the step's bookkeeping: step count, squared norm, clip coefficient,
    norm, the dropout-epoch word incremented once per step (n = 2.6 M: the bucket size of the bridge step; n = 1000: one
    workgroup)."""
    hip = _hip()
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (0.5 + i) for i in range(3)]
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=3e-3, weight_decay=0.02, betas=(0.9, 0.98), eps=1e-8)
    p, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    state = torch.zeros(8 + 1024, device="cuda")
    state[2] = 3e-3
    epoch = torch.zeros(1, dtype=torch.int32, device="cuda")
    for i, gr in enumerate(grads):
        ref.grad = gr.clone()
        norm = torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gd = gr.cuda()
        hip.call("mm_sumsq", gd, state, n)
        hip.call("mm_adamw_clip", p, gd, m, v, state, n, 0.9, 0.98, 1e-8, 0.02, 1.0, 1.0, 1, epoch)
        torch.cuda.synchronize()
        st = state[:8].cpu()
        assert st[0].item() == i + 1 and int(epoch.item()) == i + 1
        assert abs(st[4].item() - norm.item()) <= 2e-4 * norm.item()          # (fp32 sums of 2.6 M squares, two summation orders)
        assert abs(st[1].item() - norm.item() ** 2) <= 4e-4 * norm.item() ** 2
        assert abs(st[3].item() - min(1.0, 1.0 / (norm.item() + 1e-6))) <= 2e-4 * st[3].item()
        assert torch.count_nonzero(gd).item() == 0                       # zero_grad
        torch.testing.assert_close(p.cpu(), ref.detach(), rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_sample_zscore_chunked_form_vs_reference_and_the_one_workgroup_form():
    """mm_sample_zscore_bf16 (normalize_modality, run_training_lite.py:48-51, on every sample's power features): with a
    workspace, big unpadded samples (config #5: 33 x 6 272) are dealt out over 32 workgroups each (per-chunk sums in double,
    E[x^2] - mean^2) - against the fp64 reference to one bf16 step, against the one-workgroup two-pass form (no
    workspace) likewise, bit-identical between runs; small or padded samples keep the one-workgroup form either way."""
    hip = _hip()
    g = torch.Generator().manual_seed(31)
    for B, rows, chv, cht in ((3, 33, 6272, 6272), (2, 9, 100, 112), (2, 16, 64, 64)):
        x = (torch.randn(B, rows, cht, generator=g).abs() ** 3 * 40.0 + 5.0)          # heavy-tailed, far from zero mean
        x[:, :, chv:] = 0.0
        xd = x[:, :, :chv].double()
        mean = xd.mean(dim=(1, 2), keepdim=True)
        std = xd.std(dim=(1, 2), unbiased=False, keepdim=True)
        want = torch.zeros(B, rows, cht, dtype=torch.float64)
        want[:, :, :chv] = (xd - mean) / (std + 1e-8)
        outs = []
        for ws in (torch.empty(64 * B, dtype=torch.float64, device="cuda"), None,
                   torch.full((64 * B,), float("nan"), dtype=torch.float64, device="cuda")):
            out = torch.full((B, rows, cht), 7.0, dtype=torch.bfloat16, device="cuda")
            hip.call("mm_sample_zscore_bf16", x.cuda(), out, ws, B, rows, chv, cht, 1e-8)
            outs.append(out.float().cpu())
        for o in outs:
            assert ((o.double() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-4).all(), (B, rows, (o.double() - want).abs().max())
        assert torch.equal(outs[0], outs[2])
        assert ((outs[0] - outs[1]).abs() <= 2.0 ** -7 * outs[1].abs() + 1e-4).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cin", [20, 304])
def test_power_merge_modes(cin):
    """mm_power_merge (EnhancedPowerEncoder's three Conv1d(C -> 64, k = 3 | 5 | 7) + BatchNorm1d(64) branches as ONE k = 7
    layer, enhanced_models_v4.py:210-234): mode 0 == F.pad + torch.cat of the parts; mode 3's bf16 image == mm_prep_conv_weight
    of mode 0's weight, bit for bit; mode 1 hands the running statistics back and counts the batch; mode 2 adds the merged
    bias / BatchNorm / weight gradients' slices into the parts' (a null part is skipped)."""
    from multimodal_eeg_fmri_amd import ops
    hip = _hip()
    g = torch.Generator().manual_seed(cin)
    ks = (3, 5, 7)
    dev = "cuda"
    w = [torch.randn(64, cin, k, generator=g).to(dev) for k in ks]
    vecs = [[torch.randn(64, generator=g).to(dev) for _ in ks] for _ in range(5)]          # bias, gamma, beta, mean, var
    parts = (w, *vecs)
    W = torch.empty(192, cin, 7, device=dev)
    V = [torch.empty(192, device=dev) for _ in range(5)]
    ops.power_merge_call(0, parts, [W] + V, cin=cin, ks=ks)
    want_w = torch.cat([F.pad(t, ((7 - k) // 2,) * 2) for t, k in zip(w, ks)], dim=0)
    assert torch.equal(W, want_w)
    for got, trip in zip(V, vecs):
        assert torch.equal(got, torch.cat(trip))
    cinp = _cpad(cin)
    img_ref = torch.empty(192, 7, cinp, dtype=torch.bfloat16, device=dev)
    hip.call("mm_prep_conv_weight", W, img_ref, None, 192, cin, 7, cinp, 0)
    img = torch.full((192, 7, cinp), 3.0, dtype=torch.bfloat16, device=dev)
    V3 = [torch.empty(192, device=dev) for _ in range(5)]
    ops.power_merge_call(3, parts, [img] + V3, cin=cin, ks=ks, cinp=cinp)
    assert torch.equal(img.view(torch.int16), img_ref.view(torch.int16))
    assert all(torch.equal(a_, b_) for a_, b_ in zip(V3, V))
    # mode 1
    rm, rv = [torch.zeros(64, device=dev) for _ in ks], [torch.zeros(64, device=dev) for _ in ks]
    nbt = [torch.full((), 4, dtype=torch.int64, device=dev) for _ in ks]
    RM, RV = torch.randn(192, generator=g).to(dev), torch.rand(192, generator=g).to(dev)
    none3 = [None] * 3
    ops.power_merge_call(1, (none3, none3, none3, none3, rm, rv), [None, None, None, None, RM, RV], tracked=nbt, cin=cin, ks=ks)
    assert torch.equal(torch.cat(rm), RM) and torch.equal(torch.cat(rv), RV) and all(int(t) == 5 for t in nbt)
    # mode 2 (the second branch's weight and the third's bias are frozen)
    dW, dB, dG, dBe = (torch.randn(192, cin, 7, generator=g).to(dev), *(torch.randn(192, generator=g).to(dev) for _ in range(3)))
    gw = [torch.ones(64, cin, k, device=dev) for k in ks]
    gb, gg, gbe = ([torch.ones(64, device=dev) for _ in ks] for _ in range(3))
    gw2, gb2 = [gw[0], None, gw[2]], [gb[0], gb[1], None]
    ops.power_merge_call(2, (gw2, gb2, gg, gbe, none3, none3), [dW, dB, dG, dBe, None, None], cin=cin, ks=ks)
    for i, k in enumerate(ks):
        lo = (7 - k) // 2
        want_g = 1.0 + dW[64 * i:64 * i + 64, :, lo:lo + k] if i != 1 else torch.ones(64, cin, k, device=dev)
        assert torch.equal(gw[i], want_g)
        assert torch.equal(gb[i], 1.0 + dB[64 * i:64 * i + 64] if i != 2 else torch.ones(64, device=dev))
        assert torch.equal(gg[i], 1.0 + dG[64 * i:64 * i + 64]) and torch.equal(gbe[i], 1.0 + dBe[64 * i:64 * i + 64])

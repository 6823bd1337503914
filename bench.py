#!/usr/bin/env python3
"""Benchmark of the north-star hot path: one data-parallel contrastive training
step (EEG temporal encoder + fMRI voxel encoder + projection bridge + InfoNCE,
forward + backward + clip + AdamW) per "step", 32 (EEG-epoch, fMRI-volume)
pairs per GPU of synthetic 64-ch x 1024-sample EEG and 32^3-voxel fMRI
(BASELINE.json configs[1]; configs[2] at --gpus 8).

Prints ONE JSON line (rank 0).  `value` = pairs/sec of the whole job, inputs
resident in HBM.  `roofline` is for the layer-2 3-D conv implicit-GEMM kernel,
timed live with HIP events on the launch stream inside the timed steps.
`cpu_baseline` = the CPU oracle (fp32 torch restatement of the same step) on the
host cores, rank 0 at N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0          # dense, /opt/skills/guides/MI355X_MICROARCH.md
PAIRS_PER_GPU = 32
PRECONDITION_STEPS = 300                # untimed steps in front of the --warmup steps (clock conditioning, ~0.25 s)
EEG_CH, EEG_T, VOL = 64, 1024, (32, 32, 32)


PMC_SUMMARY = "profiles/r03_pmc_wres_c2.summary.txt"
PMC_SUMMARY_C4 = "profiles/r03_pmc_wres_c4.summary.txt"
# rocprofv3 --kernel-trace of this command (profiles/run_prof.sh): the kernel's mean duration inside the replayed step
PROFILE_IN_STEP = {"source": "profiles/r03_step_kernel_summary.txt", "avg_launch_ms": 0.0223}


def pmc_traffic_bytes(path=None):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC summary (None if absent)"""
    import re
    try:
        text = open(os.path.join(ROOT, path or PMC_SUMMARY)).read()
    except OSError:
        return None
    m = re.search(r"=\s*([0-9.]+)\s*MB\s*$", text, re.M)
    return float(m.group(1)) * 1e6 if m else None


def cpu_model() -> str:
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cgroup_cpu_limit():
    """CPUs the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(pairs: int, steps: int = 5, warmup: int = 2):
    """oracle/ training step on the host (checker code, reported baseline only): every core of the affinity mask,
    2 warm-up + 5 timed steps, CPU model stated."""
    import torch.nn.functional as F
    from oracle import ref_functional as RF
    import multimodal_eeg_fmri_amd.enhanced_models_v4 as E
    import multimodal_eeg_fmri_amd.fmri_utils as Fm
    import multimodal_eeg_fmri_amd.bridge_utils as Bu
    affinity = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_limit()
    # every core this process may use: the affinity mask, capped by the container's CPU quota when there is one (a GPU box
    # hands each 1-GPU job a share of the host: more threads than that share only add contention)
    cores = max(1, min(affinity, int(quota + 0.5))) if quota else affinity
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    mods = {"e.": E.EnhancedERPEncoder(EEG_CH, 128, 2, 4, 0.0), "f.": Fm.fMRIVolumeEncoder3D(1, 64, dropout=0.0),
            "h.": Bu.EEGfMRIContrastiveBridge(dropout=0.0)}
    sd = {}
    for pre, m in mods.items():
        for k, v in m.state_dict().items():
            sd[pre + k] = v.detach().clone().requires_grad_(v.is_floating_point())
    leaves = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=1e-4, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    eeg = torch.randn(pairs, EEG_CH, EEG_T, generator=g)
    vol = torch.randn(pairs, 1, *VOL, generator=g)

    def step():
        opt.zero_grad()
        fe = RF.erp_encoder(sd, eeg, "e.", train=True)
        ff = RF.volume_encoder3d(sd, vol, "f.", train=True)
        ze, zf = RF.contrastive_head(sd, fe, ff, "h.bridge.")
        loss = RF.clip_loss(ze, zf, ze, zf, sd["h.logit_scale"].exp())[0]
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in leaves if p.grad is not None], 1.0)
        opt.step()
    t0 = time.perf_counter()
    for _ in range(warmup):
        step()
    warm = (time.perf_counter() - t0) / warmup
    steps = max(2, min(steps, int(25.0 / max(warm, 1e-3))))      # bounded sample: ~10-30 s of CPU work in all
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    torch.set_num_threads(1)                  # "also report 1-thread for reference"
    t0 = time.perf_counter()
    step()
    dt1 = time.perf_counter() - t0
    torch.set_num_threads(cores)
    c1 = cpu_baseline_c1_lite()
    return {"value": pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "affinity_cores": affinity, "cgroup_cpu_limit": quota,
            "sample": f"{steps} timed + {warmup} warm-up full training steps of {pairs} pairs "
                      f"(same shapes), fp32 torch CPU oracle, {dt:.2f} s/step, torch threads = {cores} = min(affinity mask "
                      f"{affinity}, container CPU quota {quota if quota else 'none'})",
            "one_thread": {"value": pairs / dt1, "unit": "pairs/s", "sample": f"1 step, {dt1:.1f} s"},
            "c1_lite": c1}


def cpu_baseline_c1_lite(steps: int = 20):
    """BASELINE configs[0] (the reference's own CPU-runnable case): B=8, 8 ch x 256 ERP + power, 459
    connectivity features through the V4-Lite classifier; oracle training step (label-smoothed CE,
    clip 1.0, AdamW lr 5e-5 wd 0.01 as run_training_lite.py:465-488) on the host cores."""
    from oracle import ref_functional as RF
    import multimodal_eeg_fmri_amd.crossmodal_v4_enhancements as Cv
    torch.manual_seed(0)
    m = Cv.EnhancedTriModalFusionNetV4Lite(8, 8, 459, dropout=0.0)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    leaves = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=5e-5, weight_decay=0.01)
    g = torch.Generator().manual_seed(1234)
    erp, pw, conn = torch.randn(8, 8, 256, generator=g), torch.randn(8, 8, 256, generator=g), torch.randn(8, 459, generator=g)
    y = torch.arange(8) % 2

    def step():
        opt.zero_grad()
        loss = RF.label_smoothing_ce(RF.trimodal_lite(sd, erp, pw, conn, train=True)[0], y, 0.1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in leaves if p.grad is not None], 1.0)
        opt.step()
    for _ in range(2):
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return {"value": 8 / dt, "unit": "samples/s", "sample": f"{steps} timed + 2 warm-up V4-Lite training steps of 8 "
            f"samples, fp32 torch CPU oracle, {dt * 1e3:.1f} ms/step"}


def standalone_wres(B: int, D: int, H: int, W: int, launches: int = 20, rounds: int = 7):
    """the roofline kernel alone (layer 2 of the voxel encoder, 32 -> 64 channels, bf16 out + BatchNorm sums) at volume
    (B, D, H, W): `launches` back-to-back launches between two HIP events on the launch stream, behind a device-side
    sleep so that the host's enqueue time is not in the bracket; per-launch mean of each round -> min / mean / max."""
    import math
    from multimodal_eeg_fmri_amd import _hip
    Cin, Cout = 32, 64
    x = torch.randn(B, D, H, W, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 27, device="cuda") / math.sqrt(Cin * 27)
    wf = torch.empty(Cout, 27, Cin, dtype=torch.bfloat16, device="cuda")
    _hip.call("mm_prep_conv_weight", w.contiguous(), wf, None, Cout, Cin, 27, Cin, 0)
    out = torch.empty(B, D, H, W, Cout, dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(32, 2, Cout, device="cuda")
    bias = torch.randn(Cout, device="cuda")

    def fn():
        _hip.call("mm_conv3d_fwd", x, wf, B, D, H, W, Cin, Cout, bias, stats, None, out)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    per = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(4.0e6))
        a.record()
        for _ in range(launches):
            fn()
        b.record()
        torch.cuda.synchronize()
        per.append(a.elapsed_time(b) / launches)
    flops = 2.0 * B * D * H * W * Cin * Cout * 27
    mean = sum(per) / len(per)
    return {"flops_per_launch": flops, "avg_launch_ms": mean, "min_launch_ms": min(per), "max_launch_ms": max(per),
            "achieved": flops / (mean * 1e-3) / 1e12, "frac": flops / (mean * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
            "launches": launches * rounds,
            "method": f"{rounds} rounds of {launches} back-to-back launches between two HIP events on the launch stream "
                      "(launch-to-launch gaps included), inputs resident"}


def fit_and_retrieve(steps: int, lr: float = 3e-4, dropout: float = 0.1, held_out_batches: int = 8):
    """Second half of BASELINE's metric (contrastive top-1 retrieval), untimed: a fresh trainer (same
    kernels, same hipGraph step; dropout 0.1 - at the timed run's 0.3 the same number of steps only reaches
    ~0.17) is fitted on fresh synthetic pairs (every step a new batch drawn on the GPU from the
    shared-latent generator of SURVEY.md 8d), then scored on batches it has never seen."""
    from multimodal_eeg_fmri_amd import ops
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    torch.manual_seed(0)
    ops.set_dropout_seed(20260)                                  # the figure must not depend on what drew masks earlier in the process
    tr = BridgeTrainer(eeg_channels=EEG_CH, dropout=dropout, lr=lr).train()
    gm = torch.Generator().manual_seed(99)                       # the fixed mixing matrices of synthetic_pairs()
    A_e = torch.randn(EEG_CH, 16, generator=gm).cuda()
    A_f = (torch.randn(VOL[0] * VOL[1] * VOL[2], 16, generator=gm) / 4.0).cuda()
    gg = torch.Generator(device="cuda").manual_seed(7)

    def fresh():
        z = torch.randn(PAIRS_PER_GPU, 16, device="cuda", generator=gg)
        e = (z @ A_e.t()).unsqueeze(-1) * 0.5 + torch.randn(PAIRS_PER_GPU, EEG_CH, EEG_T, device="cuda", generator=gg)
        f = (z @ A_f.t()).view(PAIRS_PER_GPU, 1, *VOL) + torch.randn(PAIRS_PER_GPU, 1, *VOL, device="cuda", generator=gg)
        return e, f
    # the reference's schedule (CosineAnnealingWarmup, run_training_lite.py:176 / crossmodal_v4_enhancements.py:55-78): linear
    # warm-up over the first 5 %, cosine decay to 1 % of the base rate; the rate is a device word the captured step reads,
    # rewritten every 50 steps.  (At a constant 3e-4 the held-out figure of one 6 000-step run swung between 0.43 and 0.96
    # with the dropout seeds alone - tools/r3/fit_check3.py - i.e. it measured where the noisy tail of training happened to stop.)
    import math
    warm = max(1, steps // 20)
    t0 = time.perf_counter()
    for i in range(steps):
        if i % 50 == 0:
            f = (i + 50) / warm if i < warm else 0.01 + 0.99 * 0.5 * (1.0 + math.cos(math.pi * (i - warm) / max(1, steps - warm)))
            tr.set_lr(lr * min(1.0, f))
        out = tr.train_step(*fresh())
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    acc = [0.0, 0.0, 0.0]
    for i in range(held_out_batches):
        ev = tr.evaluate(*synthetic_pairs(PAIRS_PER_GPU, EEG_CH, EEG_T, VOL, seed=900 + i))
        acc = [acc[0] + ev["top1_e2f"].item(), acc[1] + ev["top1_f2e"].item(), acc[2] + ev["loss"].item()]
    n = float(held_out_batches)
    return {"eeg_to_fmri": acc[0] / n, "fmri_to_eeg": acc[1] / n, "loss": acc[2] / n, "chance": 1.0 / PAIRS_PER_GPU,
            "fit_steps": steps, "fit_lr": lr, "fit_schedule": "5 % linear warm-up, cosine decay to 1 %", "fit_dropout": dropout,
            "fit_seconds": secs,
            "train_loss_last": out["loss"].item(),
            "held_out_pairs": held_out_batches * PAIRS_PER_GPU,
            "data": "fresh synthetic pairs per step (shared 16-d latent + unit noise), never-seen batches for scoring"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--profile", action="store_true", help="skip the post-region event-timing steps (for rocprofv3 runs)")
    ap.add_argument("--fit-steps", type=int, default=6000, help="untimed steps on FRESH synthetic pairs after the timed "
                    "region, for the held-out top-1 retrieval figure (0 = skip; single-GPU runs only)")
    ap.add_argument("--stamps", action="store_true", help="diagnostic: in-graph phase stamps of the step (adds 13 tiny nodes; "
                    "prints a phase table to stderr, the JSON line is then not a valid benchmark)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU (RCCL).  MM_DIST_BACKEND=gloo is the one-GPU rehearsal of the N > 1 code path:
    # several ranks then share a device (LOCAL_RANK modulo the device count) and exchange over gloo.
    backend = os.environ.get("MM_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
            from tools import gloo_staging           # one-GPU rehearsal of the N > 1 path (tests only)
            gloo_staging.install()
        group = dist.group.WORLD
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from multimodal_eeg_fmri_amd import ops
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    torch.manual_seed(0)                      # identical initial weights on every rank
    tr = BridgeTrainer(eeg_channels=EEG_CH, dropout=args.dropout, group=group).train()
    eeg, fmri = synthetic_pairs(PAIRS_PER_GPU, EEG_CH, EEG_T, VOL, seed=1234 + rank)
    if args.stamps:
        tr.stamps = torch.zeros(16, dtype=torch.int64, device="cuda")

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # a fresh synthetic batch every step: NBATCH pre-drawn batches cycle through the captured step's static input
    # buffers.  `value` (the contract's figure) keeps the inputs resident in HBM: the batches sit on the device and
    # a step starts with two device-to-device copies into the static buffers.  They are drawn BEFORE the warm-up:
    # drawing them (seconds of host work) between warm-up and timed loop let the GPU fall idle, and the timed 0.2 s
    # then started on a chip still ramping its clocks (run-to-run spread of 25 %).
    NBATCH = 4
    dev_batches = [synthetic_pairs(PAIRS_PER_GPU, EEG_CH, EEG_T, VOL, seed=1234 + rank + 1000 * i) for i in range(NBATCH)]
    tr.train_step(eeg, fmri)                  # capture (not a warm-up step: lazy initialisation, graph capture)
    # clock conditioning (untimed, disclosed as `preconditioning_steps`): the process has kept the GPU nearly idle for seconds
    # (imports, capture), and the chip needs ~25 ms of this workload to settle - per-10-step event timings read 0.86, 0.84, 0.82
    # and then 0.815 ms per step (tools/r3/jitter.py).  A short --warmup would put that ramp inside a short timed region.
    precondition = 0 if args.profile else PRECONDITION_STEPS        # (profiler runs count kernels per step: keep their traces short)
    import gc
    # a generation-2 collection in the timed loop is a 20-30 ms host stall (seen 1 run in 8): collector frozen and off.
    # Done BEFORE the conditioning / warm-up steps: the collection itself stalls the host for longer than the queued
    # steps last, and a GPU that has idled >= 20 ms runs its next ~10 steps 4 % slower (tools/r3/jitter2.py: blocks of
    # 10 steps after a synchronize 0.79 ms, after synchronize + 20 ms of sleep 0.83 ms) - with --steps 20 that was half
    # the timed region.
    gc.collect()
    gc.freeze()
    gc.disable()
    for i in range(precondition):
        tr.train_step(*dev_batches[i % NBATCH])
    for i in range(args.warmup):
        tr.train_step(*dev_batches[i % NBATCH])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = tr.train_step(*dev_batches[i % NBATCH])
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    loss_timed = out["loss"].item()           # read before any other step overwrites the trainer's result buffer
    tr_capture_mode = tr.capture_mode
    t = torch.tensor([dt], device="cuda")
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()

    # the same loop with every batch coming from pinned host memory: H2D into a staging pair on a copy stream
    # (12.6 MB per step), overlapped with the previous step; reported beside `value`, never as `value`
    dt_h2d = None
    if not args.profile:
        host = [(e.cpu().pin_memory(), f.cpu().pin_memory()) for e, f in dev_batches]
        stage = [(torch.empty_like(dev_batches[0][0]), torch.empty_like(dev_batches[0][1])) for _ in range(2)]
        copy_s = torch.cuda.Stream()
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]

        def upload(i):
            b = i % 2
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(consumed[b])               # the step that read this staging pair is done with it
                stage[b][0].copy_(host[i % NBATCH][0], non_blocking=True)
                stage[b][1].copy_(host[i % NBATCH][1], non_blocking=True)
                ready[b].record(copy_s)
        for b in range(2):
            consumed[b].record()

        def h2d_loop(n):
            upload(0)
            for i in range(n):
                if i + 1 < n:
                    upload(i + 1)
                torch.cuda.current_stream().wait_event(ready[i % 2])
                tr.train_step(*stage[i % 2])
                consumed[i % 2].record()
        gc.collect()                                    # (before the warm-up: no host stall between warm-up and timed loop)
        gc.disable()
        h2d_loop(max(6, min(args.warmup, 20)))        # its own warm-up: copy stream, pinned copies, event pool (first use
        sync()                                          # of each costs milliseconds - more than a 20-step timed region)
        t0 = time.perf_counter()
        h2d_loop(args.steps)
        sync()
        dt_h2d = time.perf_counter() - t0
        gc.enable()
        t = torch.tensor([dt_h2d], device="cuda")
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_h2d = t.item()
    eeg, fmri = dev_batches[0]
    if args.stamps and rank == 0:
        acc = torch.zeros(16, dtype=torch.float64)
        for _ in range(20):
            tr.train_step(eeg, fmri)
            torch.cuda.synchronize()
            s = tr.stamps.cpu().double()
            acc += (s - s[0]) / 100.0
        for i, name in enumerate(tr.STAMP_NAMES):
            print(f"  {name:22s} {acc[i].item() / 20:8.1f} us", file=sys.stderr)

    # roofline line: the timed steps are hipGraph replays (no host code runs inside them; HIP cannot record timing
    # events from inside a replay: hipEventRecordWithFlags(external) is rejected on ROCm 7.2), so the layer-2 conv3d
    # kernel is bracketed with HIP events on its launch stream in a few extra steps of the SAME tape run eagerly right
    # after the timed region (same kernels, the other stream busy as in the real step).  Eager launches are host-bound
    # (~30 us of Python per launch against ~10 us kernels): each step is queued behind a device-side sleep, so that
    # the whole step sits in the stream queues before the GPU starts and the events see device time, not host gaps.
    kt = kt_raw = kt_pair = None
    kt_all = []
    rf_c2 = rf_c4 = None
    if not args.profile:
        tr.mode = "manual"
        tr.train_step(eeg, fmri)
        ops.kernel_timer.reset("conv3d_fwd_c32")
        ops.kernel_timer.reset("event_pair_c32")
        for _ in range(16):
            torch.cuda._sleep(int(2.0e7))                    # ~10 ms of device spin: covers the host's enqueue time
            tr.train_step(eeg, fmri)
            torch.cuda.synchronize()
        kt_all = ops.kernel_timer.all_ms("conv3d_fwd_c32")
        kt_raw = sum(kt_all) / len(kt_all)
        kt_pair = ops.kernel_timer.mean_ms("event_pair_c32")      # an empty event bracket on the same stream, same steps
        kt = kt_raw - kt_pair                                     # the kernel's share of its bracket
        if rank == 0:
            # the same kernel alone: the C2 shape, and BASELINE config #4 (64 x 64 x 48 volumes -> layer 2 at 32 x 32 x 24)
            rf_c2 = standalone_wres(PAIRS_PER_GPU, 16, 16, 16)
            rf_c4 = standalone_wres(PAIRS_PER_GPU, 32, 32, 24)
    ev = tr.evaluate(eeg, fmri)
    fit = None
    if world == 1 and args.fit_steps > 0 and not args.profile:
        fit = fit_and_retrieve(args.fit_steps)
    if rank != 0:
        return
    global_batch = PAIRS_PER_GPU * world
    # layer-2 conv3d forward: M = 32 * 16^3, N = 64, K = 27 * 32
    flops = 2.0 * PAIRS_PER_GPU * 16 ** 3 * 64 * 27 * 32
    achieved = flops / (kt * 1e-3) / 1e12 if kt else None
    line = {
        "metric": "pairs_per_sec_per_node", "value": global_batch * args.steps / dt, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preconditioning_steps": precondition,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": "C2 bridge train step: 64ch x 1024 EEG (EnhancedERPEncoder) + 32^3 fMRI "
                               "(3-D conv encoder) + projection bridge + InfoNCE, fwd+bwd+clip+AdamW",
                   "pairs_per_gpu": PAIRS_PER_GPU, "global_batch": global_batch, "parallelism": f"dp{world}",
                   "dropout": args.dropout, "mfma_operands": "bf16", "accumulate": "fp32",
                   "execution": "hipGraph replay" if world == 1 else
                   f"{tr_capture_mode} (all-gather of embeddings; all-reduce of the fMRI third of the gradient bucket beside the "
                   "EEG backward, of the remainder after it)"},
        "top1_retrieval_acc": {"eeg_to_fmri": ev["top1_e2f"].item(), "fmri_to_eeg": ev["top1_f2e"].item(),
                               "chance": 1.0 / global_batch, "note": "on the training batch after the timed steps",
                               "held_out_after_fit": fit},
        "final_loss": loss_timed,
        "value_with_input_transfer": (global_batch * args.steps / dt_h2d) if dt_h2d else None,
        "input_transfer": "every step's batch copied from pinned host memory (12.6 MB) on a copy stream into a staging pair, "
                          "overlapped with the previous step; `value` keeps the batches resident in HBM",
        "roofline": {"kernel": "conv3d_wres_kernel (layer 2: 32->64 ch @16^3, implicit GEMM M=131072 N=64 K=864)",
                     "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": (achieved / PEAK_BF16_MFMA_TFLOPS) if achieved else None,
                     "flops_per_launch": flops, "avg_launch_ms": kt,
                     "event_bracket_ms": kt_raw if kt else None, "empty_event_bracket_ms": kt_pair if kt else None,
                     # spread over the bracketed launches (each minus the mean empty bracket), and the uncorrected bound:
                     # frac_raw_bracket counts the event pair's own ~5 us as kernel time (a lower bound on the fraction)
                     "min_launch_ms": (min(kt_all) - kt_pair) if kt else None,
                     "max_launch_ms": (max(kt_all) - kt_pair) if kt else None,
                     "launches_bracketed": len(kt_all),
                     "frac_raw_bracket": (flops / (kt_raw * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS) if kt else None,
                     "profile_in_step": PROFILE_IN_STEP,
                     # HBM bytes per launch from the committed rocprofv3 PMC summary (FETCH_SIZE x2 gfx950 correction
                     # and WRITE_SIZE in separate passes, profiles/run_pmc_wres.sh); algorithmic = 8.4 + 16.8 + 0.1 MB
                     "traffic": pmc_traffic_bytes(), "traffic_source": PMC_SUMMARY, "algorithmic_bytes": 25.3e6,
                     "note": "measured inside the training step (other stream busy); stand-alone and config-#4 figures: "
                             "profiles/README.md"},
    }
    if rf_c2:
        line["roofline_c2_standalone"] = dict(rf_c2, kernel="conv3d_wres_kernel alone at the C2 shape (B=32, 16^3)", bound="mfma",
                                              peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", traffic=pmc_traffic_bytes(),
                                              traffic_source=PMC_SUMMARY, algorithmic_bytes=25.3e6)
    if rf_c4:
        line["roofline_c4"] = dict(rf_c4, kernel="conv3d_wres_kernel at BASELINE config #4: 64x64x48 volumes -> layer 2 (32->64 ch) "
                                                 "@32x32x24, B=32, implicit GEMM M=786432 N=64 K=864", bound="mfma",
                                   peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", traffic=pmc_traffic_bytes(PMC_SUMMARY_C4),
                                   traffic_source=PMC_SUMMARY_C4, algorithmic_bytes=151.1e6)
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(PAIRS_PER_GPU)
    print(json.dumps(line))


if __name__ == "__main__":
    main()

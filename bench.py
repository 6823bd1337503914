#!/usr/bin/env python3
"""Benchmark of the north-star hot path: one data-parallel contrastive training
step (EEG temporal encoder + fMRI voxel encoder + projection bridge + InfoNCE,
forward + backward + clip + AdamW) per "step", 32 (EEG-epoch, fMRI-volume)
pairs per GPU of synthetic 64-ch x 1024-sample EEG and 32^3-voxel fMRI
(BASELINE.json configs[1]; configs[2] at --gpus 8).  `--config c4` runs the same
step on 64 x 64 x 48 volumes (configs[3]), `--config c5` with the multi-scale
STFT front-end + power encoder as the EEG branch (configs[4]).

`python bench.py --gpus N` without a launcher starts its own N ranks (one per
GPU, `torch.distributed.run`) BEFORE anything touches the GPU, relays rank 0's
line and exits non-zero if a rank fails; under `torch.distributed.run` (RANK /
WORLD_SIZE set) it is one of those ranks.

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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def self_launch(n: int) -> int:
    """`bench.py --gpus N` started without a launcher: this process becomes the parent of N ranks and never touches the
    GPU (no torch import, no HIP call; nothing is re-exec'ed).  Rank 0's JSON line is relayed on stdout, everything else
    the ranks print goes to stderr; the exit code is the job's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith("{"):
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited 0 without a result line\n")
        return 1
    return rc


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _ap = argparse.ArgumentParser(add_help=False)
    _ap.add_argument("--gpus", type=int, default=1)
    _n = _ap.parse_known_args()[0].gpus
    if _n > 1:
        sys.exit(self_launch(_n))

import torch  # noqa: E402  (after the self-launch decision: the parent of a multi-rank job never loads it)

PEAK_BF16_MFMA_TFLOPS = 2500.0          # dense, /opt/skills/guides/MI355X_MICROARCH.md
PAIRS_PER_GPU = 32
PRECONDITION_STEPS = 300                # untimed steps in front of the --warmup steps (clock conditioning, ~0.25 s)
EEG_CH, EEG_T = 64, 1024
STFT = dict(n_ffts=(64, 128), hop=32)   # BASELINE config #5 / SURVEY.md 8(d)

# --config: which BASELINE configuration the step runs on (c2 is the one `metric` is quoted on; c4 / c5 are extra legs)
CONFIGS = {
    "c2": dict(vol=(32, 32, 32), eeg="erp",
               workload="C2 bridge train step: 64ch x 1024 EEG (EnhancedERPEncoder) + 32^3 fMRI (3-D conv encoder) + "
                        "projection bridge + InfoNCE, fwd+bwd+clip+AdamW"),
    "c4": dict(vol=(64, 64, 48), eeg="erp",
               workload="C4 bridge train step: 64ch x 1024 EEG (EnhancedERPEncoder) + full-resolution 64x64x48 fMRI (3-D conv "
                        "encoder) + projection bridge + InfoNCE, fwd+bwd+clip+AdamW"),
    "c5": dict(vol=(32, 32, 32), eeg="stft",
               workload="C5 bridge train step: raw 64ch x 1024 EEG -> multi-scale STFT power (n_fft 64 + 128, hop 32, z-scored) -> "
                        "EnhancedPowerEncoder + 32^3 fMRI (3-D conv encoder) + projection bridge + InfoNCE, fwd+bwd+clip+AdamW"),
}

PMC_SUMMARY = "profiles/r04_pmc_wres_c2.summary.txt"
PMC_SUMMARY_C4 = "profiles/r04_pmc_wres_c4.summary.txt"
# rocprofv3 --kernel-trace of this command (profiles/run_prof.sh): per-kernel mean durations inside the replayed step
PROFILE_STEP_SUMMARIES = ("profiles/r04_step_kernel_summary.txt", "profiles/r03_step_kernel_summary.txt")


def conv3_flops(cin, cout, vox):
    return 2.0 * cin * cout * 27 * vox


def family_flops(vol, pairs):
    """the six dense 3-D convolution GEMMs of one step (layer 2 at vol/2, layer 3 at vol/4: forward, data gradient,
    weight gradient): name -> (FLOPs per launch, kernel-summary pattern (name prefix, grid))"""
    v2 = pairs * (vol[0] // 2) * (vol[1] // 2) * (vol[2] // 2)
    v3 = v2 // 8
    l2, l3 = conv3_flops(32, 64, v2), conv3_flops(64, 128, v3)
    return {"L2 fwd": l2, "L3 fwd": l3, "L3 dgrad": l3, "L2 dgrad": l2, "L2 wgrad": l2, "L3 wgrad": l3}


def step_flops(cfg, pairs):
    """algorithmic FLOPs of one training step (forward per pair, backward = 2 x forward)"""
    T, C, d, ff = EEG_T, EEG_CH, 128, 512

    def blocks(L):
        return 2 * (2 * L * d * 3 * d + 4 * L * L * d + 2 * L * d * d + 2 * 2 * L * d * ff)
    if cfg["eeg"] == "erp":
        eeg = 2 * T * 64 * C * 7 + 2 * T * 128 * 64 * 5 + 2 * (T // 2) * d * 128 * 3 + blocks(T // 2) + 2 * d * d
    else:
        frames = T // STFT["hop"] + 1
        cs = C * sum(n // 2 + 1 for n in STFT["n_ffts"])
        eeg = 2 * frames * 64 * cs * (3 + 5 + 7) + 2 * frames * d * 192 + blocks(frames) + 2 * d * d
    v = cfg["vol"][0] * cfg["vol"][1] * cfg["vol"][2]
    vox = conv3_flops(1, 32, v) + conv3_flops(32, 64, v // 8) + conv3_flops(64, 128, v // 64) + 2 * 128 * 64
    return 3.0 * pairs * (eeg + vox)


def profile_family_us(vol=(32, 32, 32)):
    """mean in-step durations (us) of the family's six launches parsed from the committed rocprofv3 kernel table
    (profiles/summarize.py output); None when no summary is committed or a row is missing"""
    import re
    if tuple(vol) != (32, 32, 32):
        return None, None
    pats = {"L2 fwd": r"conv3d_wres_kernel\s", "L3 fwd": r"conv3d_stream_kernel<64, 128,", "L3 dgrad": r"conv3d_stream_kernel<128, 64,",
            "L2 dgrad": r"conv3d_stream_kernel<64, 32,", "L2 wgrad": r"conv3d_wgrad_kernel\s+grid=\(41,", "L3 wgrad": r"conv3d_wgrad_kernel\s+grid=\(10,"}
    for path in PROFILE_STEP_SUMMARIES:
        try:
            text = open(os.path.join(ROOT, path)).read()
        except OSError:
            continue
        out = {}
        for k, pat in pats.items():
            m = re.search(pat + r".*?avg=\s*([0-9.]+)us", text)
            if m:
                out[k] = float(m.group(1))
        if len(out) == len(pats):
            return out, path
    return None, None


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


def make_eeg_encoder(kind: str, dropout: float):
    """the EEG branch of a --config (None = the trainer's default EnhancedERPEncoder)"""
    if kind == "erp":
        return None
    from multimodal_eeg_fmri_amd.crossmodal_v4_enhancements import MultiScaleSTFTPowerEncoder
    return MultiScaleSTFTPowerEncoder(EEG_CH, STFT["n_ffts"], STFT["hop"], 128, 2, 4, dropout)


def cpu_baseline(pairs: int, cfg, steps: int = 5, warmup: int = 2):
    """oracle/ training step on the host (checker code, reported baseline only): every core of the affinity mask,
    2 warm-up + 5 timed steps, CPU model stated."""
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
    enc = make_eeg_encoder(cfg["eeg"], 0.0) or E.EnhancedERPEncoder(EEG_CH, 128, 2, 4, 0.0)
    mods = {"e.": enc, "f.": Fm.fMRIVolumeEncoder3D(1, 64, dropout=0.0), "h.": Bu.EEGfMRIContrastiveBridge(dropout=0.0)}
    sd = {}
    for pre, m in mods.items():
        for k, v in m.state_dict().items():
            sd[pre + k] = v.detach().clone().requires_grad_(v.is_floating_point())
    leaves = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.AdamW(leaves, lr=1e-4, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    eeg = torch.randn(pairs, EEG_CH, EEG_T, generator=g)
    vol = torch.randn(pairs, 1, *cfg["vol"], generator=g)

    def step():
        opt.zero_grad()
        if cfg["eeg"] == "erp":
            fe = RF.erp_encoder(sd, eeg, "e.", train=True)
        else:
            fe = RF.stft_power_encoder(sd, eeg, STFT["n_ffts"], STFT["hop"], "e.encoder.", train=True)
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
    out = {"value": pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
           "affinity_cores": affinity, "cgroup_cpu_limit": quota,
           "sample": f"{steps} timed + {warmup} warm-up full training steps of {pairs} pairs "
                     f"(same shapes), fp32 torch CPU oracle, {dt:.2f} s/step, torch threads = {cores} = min(affinity mask "
                     f"{affinity}, container CPU quota {quota if quota else 'none'})",
           "one_thread": {"value": pairs / dt1, "unit": "pairs/s", "sample": f"1 step, {dt1:.1f} s"}}
    if cfg is CONFIGS["c2"]:
        out["c1_lite"] = cpu_baseline_c1_lite()
    return out


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


def standalone_wres(B: int, D: int, H: int, W: int, min_launches: int = 200, min_ms: float = 15.0, rounds: int = 5):
    """the roofline kernel alone (layer 2 of the voxel encoder, 32 -> 64 channels, bf16 out + BatchNorm sums) at volume
    (B, D, H, W): per round at least `min_launches` back-to-back launches AND at least `min_ms` of device time between two
    HIP events on the launch stream (sustained clocks, not a burst - the rocprofv3 trace mean is the figure to agree with),
    behind a device-side sleep so that the host's enqueue time is not in the bracket; per-launch mean of each round."""
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

    def round_of(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(4.0e6))
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    launches = max(min_launches, int(min_ms / max(round_of(50), 1e-4)) + 1)
    per = [round_of(launches) for _ in range(rounds)]
    flops = conv3_flops(Cin, Cout, B * D * H * W)
    mean = sum(per) / len(per)
    return {"flops_per_launch": flops, "avg_launch_ms": mean, "min_launch_ms": min(per), "max_launch_ms": max(per),
            "achieved": flops / (mean * 1e-3) / 1e12, "frac": flops / (mean * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
            "launches": launches * rounds,
            "method": f"{rounds} rounds of {launches} back-to-back launches ({launches * mean:.1f} ms each) between two HIP events "
                      "on the launch stream (launch-to-launch gaps included), inputs resident"}


def voxel_encoder_only(vol, pairs: int, dropout: float, iters: int = 20):
    """BASELINE config #4 as it is worded - the 3-D conv encoder alone: forward + backward of fMRIVolumeEncoder3D through
    its public autograd surface on `pairs` volumes, HIP events around `iters` iterations behind a device-side sleep"""
    import multimodal_eeg_fmri_amd.fmri_utils as Fm
    torch.manual_seed(0)
    enc = Fm.fMRIVolumeEncoder3D(1, 64, dropout=dropout).cuda().train()
    x = torch.randn(pairs, 1, *vol, device="cuda")
    dout = torch.randn(pairs, 64, device="cuda")
    for _ in range(3):
        enc(x).backward(dout)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2.0e7))
    a.record()
    for _ in range(iters):
        enc(x).backward(dout)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    v = vol[0] * vol[1] * vol[2]
    fl = 3.0 * pairs * (conv3_flops(1, 32, v) + conv3_flops(32, 64, v // 8) + conv3_flops(64, 128, v // 64))
    return {"volumes_per_s": pairs / (ms * 1e-3), "ms_per_iteration": ms, "iterations": iters, "pairs": pairs,
            "mfma_frac_of_peak": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, "flops_per_iteration": fl,
            "what": "fMRIVolumeEncoder3D forward + backward (eager, public autograd surface), no optimizer"}


def fit_and_retrieve(steps: int, cfg, lr: float = 3e-4, dropout: float = 0.1, held_out_batches: int = 8):
    """Second half of BASELINE's metric (contrastive top-1 retrieval), untimed: a fresh trainer (same
    kernels, same hipGraph step; dropout 0.1 - at the timed run's 0.3 the same number of steps only reaches
    ~0.17) is fitted on fresh synthetic pairs (every step a new batch drawn on the GPU from the
    shared-latent generator of SURVEY.md 8d), then scored on batches it has never seen."""
    from multimodal_eeg_fmri_amd import ops
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    VOL = cfg["vol"]
    torch.manual_seed(0)
    ops.set_dropout_seed(20260)                                  # the figure must not depend on what drew masks earlier in the process
    tr = BridgeTrainer(eeg_channels=EEG_CH, dropout=dropout, lr=lr, eeg_encoder=make_eeg_encoder(cfg["eeg"], dropout)).train()
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
    # with the dropout seeds alone - profiles/scripts/r3/fit_check3.py - i.e. it measured where the noisy tail of training happened to stop.)
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
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE configuration of the step: c2 (the one the "
                    "metric is quoted on), c4 (64x64x48 volumes), c5 (multi-scale STFT front-end + power encoder as the EEG branch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.3)
    ap.add_argument("--profile", action="store_true", help="skip the post-region event-timing steps (for rocprofv3 runs)")
    ap.add_argument("--fit-steps", type=int, default=None, help="untimed steps on FRESH synthetic pairs after the timed "
                    "region, for the held-out top-1 retrieval figure (0 = skip; single-GPU runs only; default 6000 for c2, 0 otherwise)")
    ap.add_argument("--stamps", action="store_true", help="diagnostic: in-graph phase stamps of the step (adds 13 tiny nodes; "
                    "prints a phase table to stderr, the JSON line is then not a valid benchmark)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    VOL = cfg["vol"]
    if args.fit_steps is None:
        args.fit_steps = 6000 if args.config == "c2" else 0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}: start it as `python bench.py --gpus "
                         f"{args.gpus}` (it launches its own ranks) or under torch.distributed.run with --nproc-per-node {args.gpus}")
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

    from multimodal_eeg_fmri_amd import ops
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    torch.manual_seed(0)                      # identical initial weights on every rank
    tr = BridgeTrainer(eeg_channels=EEG_CH, dropout=args.dropout, group=group,
                       eeg_encoder=make_eeg_encoder(cfg["eeg"], args.dropout)).train()
    eeg, fmri = synthetic_pairs(PAIRS_PER_GPU, EEG_CH, EEG_T, VOL, seed=1234 + rank)
    if args.stamps:
        tr.stamps = torch.zeros(16, dtype=torch.int64, device="cuda")

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], device="cuda", dtype=torch.float64)
        if world > 1:
            import torch.distributed as dist
            if backend == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            else:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX)
                t = h
        return t.item()

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
    # and then 0.815 ms per step (profiles/scripts/r3/jitter.py).  A short --warmup would put that ramp inside a short timed region.
    precondition = 0 if args.profile else PRECONDITION_STEPS        # (profiler runs count kernels per step: keep their traces short)
    import gc
    # a generation-2 collection in the timed loop is a 20-30 ms host stall (seen 1 run in 8): collector frozen and off.
    # Done BEFORE the conditioning / warm-up steps: the collection itself stalls the host for longer than the queued
    # steps last, and a GPU that has idled >= 20 ms runs its next ~10 steps 4 % slower (profiles/scripts/r3/jitter2.py: blocks of
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
    dt = max_over_ranks(dt)

    # the same loop fed from the host (a real training loop's rate; reported beside `value`, never as `value`)
    h2d = None
    if not args.profile:
        gc.collect()                                    # (before the warm-up: no host stall between warm-up and timed loop)
        gc.disable()
        # its own conditioning: building the pinned buffers leaves the GPU idle for tens of milliseconds, and a chip that has
        # idled runs its next steps slower (see above) - with --steps 20 that was the whole timed region
        h2d = host_fed_loop(tr, dev_batches, args.steps, max(60, args.warmup), sync)
        gc.enable()
        h2d["seconds"] = max_over_ranks(h2d["seconds"])
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
    # events from inside a replay: hipEventRecordWithFlags(external) is rejected on ROCm 7.2), so the 3-D convolution
    # GEMMs are bracketed with HIP events on their launch stream in a few extra steps of the SAME tape run eagerly right
    # after the timed region (same kernels, the other stream busy as in the real step).  Eager launches are host-bound
    # (~30 us of Python per launch against ~10 us kernels): each step is queued behind a device-side sleep, so that
    # the whole step sits in the stream queues before the GPU starts and the events see device time, not host gaps.
    # A bracket also holds the event pair's own ~5 us: `frac` is that UNCORRECTED figure (a lower bound, and what the
    # rocprofv3 trace of the same command agrees with); the figure with an empty bracket subtracted is secondary.
    FAMILY = (("L2 fwd", "conv3d_fwd_c32"), ("L3 fwd", "conv3d_fwd_c64"), ("L3 dgrad", "conv3d_dgrad_c128"),
              ("L2 dgrad", "conv3d_dgrad_c64"), ("L2 wgrad", "conv3d_wgrad_c32"), ("L3 wgrad", "conv3d_wgrad_c64"))
    kt_raw = kt_pair = None
    kt_all = []
    fam_ms = {}
    rf_c2 = rf_c4 = enc_only = None
    if not args.profile:
        tr.mode = "manual"
        tr.train_step(eeg, fmri)
        for _, key in FAMILY:
            ops.kernel_timer.reset(key)
        ops.kernel_timer.reset("event_pair_c32")
        for _ in range(16):
            torch.cuda._sleep(int(2.0e7))                    # ~10 ms of device spin: covers the host's enqueue time
            tr.train_step(eeg, fmri)
            torch.cuda.synchronize()
        kt_all = ops.kernel_timer.all_ms("conv3d_fwd_c32")
        kt_raw = sum(kt_all) / len(kt_all)
        kt_pair = ops.kernel_timer.mean_ms("event_pair_c32")      # an empty event bracket on the same stream, same steps
        fam_ms = {name: ops.kernel_timer.mean_ms(key) for name, key in FAMILY}
        if rank == 0:
            # the same kernel alone: the C2 shape, and BASELINE config #4 (64 x 64 x 48 volumes -> layer 2 at 32 x 32 x 24)
            rf_c2 = standalone_wres(PAIRS_PER_GPU, 16, 16, 16)
            rf_c4 = standalone_wres(PAIRS_PER_GPU, 32, 32, 24)
            if args.config == "c4":
                enc_only = voxel_encoder_only(VOL, PAIRS_PER_GPU, args.dropout)
    coll_us = tr.time_collectives(PAIRS_PER_GPU) if world > 1 else None          # collective: every rank takes part
    ev = tr.evaluate(eeg, fmri)
    fit = None
    if world == 1 and args.fit_steps > 0 and not args.profile:
        fit = fit_and_retrieve(args.fit_steps, cfg)
    if rank != 0:
        return
    global_batch = PAIRS_PER_GPU * world
    ms_per_step = dt / args.steps * 1e3
    # layer-2 conv3d forward: M = 32 * (vol / 2)^3 voxels, N = 64, K = 27 * 32
    fam_fl = family_flops(VOL, PAIRS_PER_GPU)
    flops = fam_fl["L2 fwd"]
    v2 = tuple(v // 2 for v in VOL)

    def tf(fl, ms):
        return fl / (ms * 1e-3) / 1e12 if ms else None
    achieved = tf(flops, kt_raw)
    if world == 1:
        execution = "hipGraph replay (one graph, two streams)"
    elif tr_capture_mode and tr_capture_mode.startswith("one graph"):
        execution = (f"{tr_capture_mode}: all-gather of embeddings; all-reduce of the gradient bucket per finished layer group "
                     f"({', '.join(g[0] for g in reversed(tr.groups))}), all but the last beside the EEG backward")
    else:
        execution = f"{tr_capture_mode}: all-gather of embeddings after the forward segment; ONE all-reduce of the whole bucket after the backward segment"
    sfl = step_flops(cfg, PAIRS_PER_GPU)
    line = {
        "metric": "pairs_per_sec_per_node", "value": global_batch * args.steps / dt, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preconditioning_steps": precondition,
        "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": cfg["workload"], "baseline_config": args.config, "volume": list(VOL),
                   "pairs_per_gpu": PAIRS_PER_GPU, "global_batch": global_batch, "parallelism": f"dp{world}",
                   "dropout": args.dropout, "mfma_operands": "bf16", "accumulate": "fp32", "execution": execution},
        "top1_retrieval_acc": {"eeg_to_fmri": ev["top1_e2f"].item(), "fmri_to_eeg": ev["top1_f2e"].item(),
                               "chance": 1.0 / global_batch, "note": "on the training batch after the timed steps",
                               "held_out_after_fit": fit},
        "final_loss": loss_timed,
        # the whole step against the MFMA roof: algorithmic FLOPs of one step (SURVEY.md 8(d) formulas) / step time / 2.5 PF
        "step_mfma_frac": sfl / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, "step_flops": sfl,
        "value_with_input_transfer": (global_batch * args.steps / h2d["seconds"]) if h2d else None,
        "input_transfer": h2d,
        "roofline": {"kernel": f"conv3d_wres_kernel (layer 2: 32->64 ch @{v2[0]}x{v2[1]}x{v2[2]}, implicit GEMM "
                               f"M={PAIRS_PER_GPU * v2[0] * v2[1] * v2[2]} N=64 K=864)",
                     "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": (achieved / PEAK_BF16_MFMA_TFLOPS) if achieved else None,
                     "flops_per_launch": flops, "avg_launch_ms": kt_raw,
                     "min_launch_ms": min(kt_all) if kt_all else None, "max_launch_ms": max(kt_all) if kt_all else None,
                     "launches_bracketed": len(kt_all),
                     # secondary: the same brackets minus the mean EMPTY bracket (what a pair of event records costs alone)
                     "empty_event_bracket_ms": kt_pair,
                     "frac_minus_empty_bracket": (tf(flops, kt_raw - kt_pair) / PEAK_BF16_MFMA_TFLOPS) if kt_raw else None,
                     # HBM bytes per launch from the committed rocprofv3 PMC summary (FETCH_SIZE x2 gfx950 correction
                     # and WRITE_SIZE in separate passes, profiles/run_pmc_wres.sh); algorithmic = 8.4 + 16.8 + 0.1 MB
                     "traffic": pmc_traffic_bytes() if args.config != "c4" else pmc_traffic_bytes(PMC_SUMMARY_C4),
                     "traffic_source": PMC_SUMMARY if args.config != "c4" else PMC_SUMMARY_C4,
                     "algorithmic_bytes": 25.3e6 if args.config != "c4" else 151.1e6,
                     "note": "measured inside the training step (other stream busy), uncorrected event bracket; stand-alone and "
                             "config-#4 figures: roofline_c2_standalone / roofline_c4"},
    }
    prof1, prof1_src = profile_family_us(VOL)
    if prof1:
        # the same kernel's mean duration inside the replayed step under rocprofv3 --kernel-trace (committed table)
        line["roofline"]["profile_in_step"] = {"source": prof1_src, "avg_launch_us": prof1["L2 fwd"],
                                               "frac": flops / (prof1["L2 fwd"] * 1e-6) / 1e12 / PEAK_BF16_MFMA_TFLOPS}
    if fam_ms and all(fam_ms.values()):
        tot_fl, tot_ms = sum(fam_fl.values()), sum(fam_ms.values())
        prof, prof_src = profile_family_us(VOL)
        line["roofline_family"] = {
            "what": "the six dense 3-D convolution GEMM launches of one step (layers 2 and 3: forward, data gradient, weight "
                    "gradient), event-bracketed inside the training step like `roofline` (uncorrected)",
            "bound": "mfma", "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "flops": tot_fl, "sum_launch_ms": tot_ms,
            "achieved": tf(tot_fl, tot_ms), "frac": tf(tot_fl, tot_ms) / PEAK_BF16_MFMA_TFLOPS,
            "frac_minus_empty_bracket": tf(tot_fl, tot_ms - 6 * kt_pair) / PEAK_BF16_MFMA_TFLOPS,
            "per_launch": {k: {"ms": fam_ms[k], "flops": fam_fl[k], "frac": tf(fam_fl[k], fam_ms[k]) / PEAK_BF16_MFMA_TFLOPS}
                           for k in fam_fl},
            "profile_in_step": None if prof is None else {
                "source": prof_src, "avg_launch_us": prof,
                "frac": tot_fl / (sum(prof.values()) * 1e-6) / 1e12 / PEAK_BF16_MFMA_TFLOPS}}
    if rf_c2:
        line["roofline_c2_standalone"] = dict(rf_c2, kernel="conv3d_wres_kernel alone at the C2 shape (B=32, 16^3)", bound="mfma",
                                              peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", traffic=pmc_traffic_bytes(),
                                              traffic_source=PMC_SUMMARY, algorithmic_bytes=25.3e6)
    if rf_c4:
        line["roofline_c4"] = dict(rf_c4, kernel="conv3d_wres_kernel at BASELINE config #4: 64x64x48 volumes -> layer 2 (32->64 ch) "
                                                 "@32x32x24, B=32, implicit GEMM M=786432 N=64 K=864", bound="mfma",
                                   peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", traffic=pmc_traffic_bytes(PMC_SUMMARY_C4),
                                   traffic_source=PMC_SUMMARY_C4, algorithmic_bytes=151.1e6)
    if enc_only:
        line["voxel_encoder_only"] = enc_only
    if world > 1:
        line["rccl_ranks"] = world if backend == "nccl" else 0
        line["dist_backend"] = "RCCL (torch.distributed nccl)" if backend == "nccl" else f"{backend} (one-GPU rehearsal: ranks share a device)"
        line["capture_mode"] = tr_capture_mode
        line["gradient_bucket_groups"] = [{"name": n, "ready": r, "MB": (hi - lo) * 4 / 1e6} for n, r, lo, hi in tr.groups]
        line["collectives_us"] = coll_us
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(PAIRS_PER_GPU, cfg)
    print(json.dumps(line), flush=True)


def host_fed_loop(tr, dev_batches, steps, warm, sync):
    """every step's batch comes from pinned host memory, the way a training loop feeds the step: the loader hands over ONE
    packed pinned buffer per batch (BridgeTrainer.pack_host_batch: the EEG epochs already in the first convolution's bf16
    operand layout + the fp32 volumes - 8.4 MB instead of 12.6 MB at C2), `HostFeeder` copies it on a copy stream into a
    ring of three staging buffers, overlapped with the previous step, and orders copies and steps from the HOST (no
    cross-queue waits: they cost 8 %, profiles/r04_h2d_probe.txt); one device-to-device copy into the step's static inputs.
    -> {seconds, description}"""
    NB = len(dev_batches)
    packed = [tr.pack_host_batch(e, f) for e, f in dev_batches]        # (a loader's work, done off the timed loop)
    feeder = tr.host_feeder()

    def loop(n):
        feeder.upload(packed[0])
        for i in range(n):
            if i + 1 < n:
                feeder.upload(packed[(i + 1) % NB])
            feeder.step()
    loop(warm)            # its own warm-up: copy stream, pinned copies, event pool (first use of each costs milliseconds)
    sync()
    t0 = time.perf_counter()
    loop(steps)
    sync()
    return {"seconds": time.perf_counter() - t0, "bytes_per_step": packed[0].numel(),
            "how": "every step's batch as ONE packed pinned buffer (EEG epochs in the first convolution's bf16 operand layout + fp32 "
                   "volumes), one H2D copy on a copy stream into a ring of three staging buffers overlapped with the previous step "
                   "(copies and steps ordered by host-side event waits: BridgeTrainer.host_feeder), one D2D copy into the step's "
                   "static inputs; `value` keeps the batches resident in HBM"}


if __name__ == "__main__":
    main()

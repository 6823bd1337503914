"""per-10-step GPU event durations of the graph-replayed training step: is a slow run uniformly slow (clock state) or stalled?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
torch.manual_seed(0)
tr = BridgeTrainer(eeg_channels=64, dropout=0.3).train()
batches = [synthetic_pairs(32, 64, 1024, (32, 32, 32), seed=1234 + 1000 * i) for i in range(4)]
tr.train_step(*batches[0])
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n // 10 + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(n):
    tr.train_step(*batches[i % 4])
    if (i + 1) % 10 == 0:
        ev[(i + 1) // 10].record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
d = [ev[k].elapsed_time(ev[k + 1]) / 10 for k in range(n // 10)]
print(f"wall {wall / n * 1e3:.4f} ms/step; per-10-step GPU ms/step: first 5 {[round(x, 3) for x in d[:5]]} min {min(d):.3f} median {sorted(d)[len(d) // 2]:.3f} max {max(d):.3f}; slow blocks (>1.1x median) {[ (k, round(x, 3)) for k, x in enumerate(d) if x > 1.1 * sorted(d)[len(d) // 2]]}")

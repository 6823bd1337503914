"""what makes the first ~10 replayed steps after a synchronize slower: idle time, or the synchronize itself?
blocks of 10 steps (GPU event timing), with different things between the blocks"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
torch.manual_seed(0)
tr = BridgeTrainer(eeg_channels=64, dropout=0.3).train()
batches = [synthetic_pairs(32, 64, 1024, (32, 32, 32), seed=1234 + 1000 * i) for i in range(4)]
tr.train_step(*batches[0])
for i in range(300):
    tr.train_step(*batches[i % 4])
torch.cuda.synchronize()

def blocks(nb, between):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * nb)]
    for b in range(nb):
        between()
        ev[2 * b].record()
        for i in range(10):
            tr.train_step(*batches[i % 4])
        ev[2 * b + 1].record()
    torch.cuda.synchronize()
    return [round(ev[2 * b].elapsed_time(ev[2 * b + 1]) / 10, 4) for b in range(nb)]

print("no gap         ", blocks(8, lambda: None))
print("synchronize    ", blocks(8, torch.cuda.synchronize))
print("sync + 1 ms    ", blocks(8, lambda: (torch.cuda.synchronize(), time.sleep(0.001))))
print("sync + 20 ms   ", blocks(8, lambda: (torch.cuda.synchronize(), time.sleep(0.02))))
print("sync + 200 ms  ", blocks(5, lambda: (torch.cuda.synchronize(), time.sleep(0.2))))
print("no gap again   ", blocks(8, lambda: None))
# 20-step blocks after a synchronize: what the driver's --steps 20 sees
def blocks20(nb):
    out = []
    for b in range(nb):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            tr.train_step(*batches[i % 4])
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
    return out
print("wall, 20 steps between synchronizes", blocks20(8))

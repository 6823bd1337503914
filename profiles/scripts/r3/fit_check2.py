"""does what ran earlier in the process change the bench's fit?  argv: stages to run before the fit"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from multimodal_eeg_fmri_amd import ops
from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
stages = sys.argv[1:]
tr = None
if "graph" in stages or "manual" in stages or "timer" in stages or "eval" in stages:
    torch.manual_seed(0)
    tr = BridgeTrainer(eeg_channels=64, dropout=0.3).train()
    e, f = synthetic_pairs(32, 64, 1024, (32, 32, 32), seed=1234)
    for _ in range(30):
        tr.train_step(e, f)
if "manual" in stages or "timer" in stages:
    tr.mode = "manual"
    tr.train_step(e, f)
    if "timer" in stages:
        ops.kernel_timer.reset("conv3d_fwd_c32")
        ops.kernel_timer.reset("event_pair_c32")
    for _ in range(4):
        tr.train_step(e, f)
        torch.cuda.synchronize()
if "eval" in stages:
    tr.evaluate(e, f)
if "del" in stages:
    del tr
    import gc; gc.collect(); torch.cuda.empty_cache()
r = bench.fit_and_retrieve(6000)
print(stages, json.dumps({k: round(r[k], 4) for k in ("eeg_to_fmri", "fmri_to_eeg", "loss", "train_loss_last")}))

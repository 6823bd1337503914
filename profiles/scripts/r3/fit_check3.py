"""spread of the bench's fit result over dropout-seed offsets (fresh process each)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from multimodal_eeg_fmri_amd import ops
off = int(sys.argv[1])
_orig = ops.set_dropout_seed
ops.set_dropout_seed = lambda s: (_orig(s), ops._seed_state.__setitem__("step", off))      # the fit seeds itself: shift its draws
r = bench.fit_and_retrieve(int(sys.argv[2]) if len(sys.argv) > 2 else 6000)
print("seed-step offset", off, json.dumps({k: round(r[k], 4) for k in ("eeg_to_fmri", "fmri_to_eeg", "loss", "train_loss_last")}))

"""held-out retrieval after the bench's fit, under the A/B knobs given in the environment"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
r = bench.fit_and_retrieve(int(sys.argv[1]) if len(sys.argv) > 1 else 6000)
print(json.dumps({k: r[k] for k in ("eeg_to_fmri", "fmri_to_eeg", "loss", "train_loss_last", "fit_seconds")}))

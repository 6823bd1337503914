"""does the bridge learn the shared latent from FRESH synthetic batches? (held-out top-1 retrieval)"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
torch.manual_seed(0)
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 800
drop = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
tr = BridgeTrainer(eeg_channels=64, dropout=drop, lr=lr).train()
held = synthetic_pairs(32, 64, 1024, (32, 32, 32), seed=999)
gm = torch.Generator().manual_seed(99)
A_e = torch.randn(64, 16, generator=gm).cuda()
A_f = (torch.randn(32 ** 3, 16, generator=gm) / 4).cuda()
gg = torch.Generator(device="cuda").manual_seed(7)
def fresh():
    z = torch.randn(32, 16, device="cuda", generator=gg)
    eeg = (z @ A_e.t()).unsqueeze(-1) * 0.5 + torch.randn(32, 64, 1024, device="cuda", generator=gg)
    fmri = (z @ A_f.t()).view(32, 1, 32, 32, 32) + torch.randn(32, 1, 32, 32, 32, device="cuda", generator=gg)
    return eeg, fmri
t0 = time.time()
for i in range(steps):
    eeg, fmri = fresh()
    out = tr.train_step(eeg, fmri)
    if (i + 1) % 250 == 0:
        ev = tr.evaluate(*held)
        print(i + 1, "train loss %.3f" % out["loss"].item(), "held-out loss %.3f top1 e2f %.3f f2e %.3f" % (ev["loss"].item(), ev["top1_e2f"].item(), ev["top1_f2e"].item()), "%.1fs" % (time.time() - t0), flush=True)

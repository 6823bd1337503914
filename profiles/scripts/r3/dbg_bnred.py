"""diagnostic: is mm_conv1d_dgrad_bn_reduce deterministic launch to launch, and which trainer mode loses bit-reproducibility"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodal_eeg_fmri_amd import _hip as hip
hip.load()

def kernel_case(B, T, Cin, Cout, k, pool, p):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cin, Cout, k, generator=g) / math.sqrt(Cout * k)
    wf = torch.empty(Cin, k, Cout, dtype=torch.bfloat16, device="cuda")
    wd = torch.empty(Cout, k, Cin, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_prep_conv_weight", w.cuda().contiguous(), wf, wd, Cin, Cout, k, Cout, Cin)
    dy = (torch.randn(B, T, Cin, generator=g) * 0.1).cuda().to(torch.bfloat16)
    yb = (torch.randn(B, T * pool, Cout, generator=g) * 1.2 + 0.1).cuda()
    out4 = torch.stack([0.5 + torch.rand(Cout, generator=g), torch.randn(Cout, generator=g) * 0.2,
                        torch.randn(Cout, generator=g) * 0.1, 0.8 + 0.4 * torch.rand(Cout, generator=g)]).cuda().contiguous()
    outs = []
    for i in range(6):
        dx = torch.empty(B, T, Cout, dtype=torch.bfloat16, device="cuda")
        sums = torch.zeros(32, 2, Cout, device="cuda")
        hip.call("mm_conv1d_dgrad_bn_reduce", dy, wd, B, T, Cin, Cout, k, k - 1 - k // 2, dx, yb, out4, sums, 1, pool, 0, p, 99, None)
        torch.cuda.synchronize()
        outs.append((dx.clone(), sums.view(torch.int32).clone()))
    ok = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    print("kernel", (B, T, Cin, Cout, k, pool, p), "deterministic" if ok else "DIFFERS",
          [int((outs[0][1] != o[1]).sum()) for o in outs[1:]])

kernel_case(32, 512, 128, 128, 3, 2, 0.1)
kernel_case(32, 1024, 128, 64, 5, 1, 0.1)

from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
from multimodal_eeg_fmri_amd import ops

def trainer(mode):
    Bsz, C, T, vol = 32, 64, 1024, (32, 32, 32)
    batches = [synthetic_pairs(Bsz, C, T, vol, seed=100 + i) for i in range(3)]
    def run():
        ops.set_seed_epoch(None); ops.set_dropout_seed(1234); torch.manual_seed(0)
        tr = BridgeTrainer(eeg_channels=C, dropout=0.1, lr=1e-3, mode=mode).train()
        losses = []
        for i in range(6):
            eeg, fmri = batches[i % 3]
            losses.append(tr.train_step(eeg, fmri)["loss"].clone())
        torch.cuda.synchronize()
        ops.set_seed_epoch(None)
        return torch.stack(losses)
    a, b = run(), run()
    print("trainer", mode, "bnred off" if os.environ.get("MM_NO_BNRED") else "bnred on", "equal" if torch.equal(a, b) else f"DIFFERS {(a-b).abs().max().item():.3e}", a.tolist()[:3])

for m in sys.argv[1:]:
    trainer(m)

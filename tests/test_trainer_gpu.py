"""GPU: contrastive bridge head / loss parity and one-GPU trainer behaviour."""
import pytest
import torch

from oracle import ref_functional as RF
from oracle.fixtures import build, seeded_randn

import multimodal_eeg_fmri_amd.bridge_utils as B

pytestmark = pytest.mark.gpu


def test_contrastive_head_and_loss_match_oracle():
    """a-X2 (extension; parity unpinned by reference): embeddings, loss, top-1 and
    all gradients vs the CPU restatement, fp32 kernels -> 1e-4 tolerances."""
    m = build(B.EEGfMRIContrastiveBridge, 41, dropout=0.0).train()
    eeg, fmri = seeded_randn(141, 16, 128), seeded_randn(142, 16, 64)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    eo, fo = eeg.clone().requires_grad_(True), fmri.clone().requires_grad_(True)
    ze, zf = RF.contrastive_head(sd, eo, fo, "bridge.")
    loss, ae, af, _ = RF.clip_loss(ze, zf, ze, zf, sd["logit_scale"].exp())
    loss.backward()
    mg = m.cuda()
    eg, fg = eeg.cuda().requires_grad_(True), fmri.cuda().requires_grad_(True)
    ge, gf = mg.embed(eg, fg)
    torch.testing.assert_close(ge.detach().cpu(), ze.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gf.detach().cpu(), zf.detach(), rtol=1e-4, atol=1e-5)
    l2, a2e, a2f = mg(eg, fg)
    l2.backward()
    assert abs(l2.item() - loss.item()) < 1e-4
    assert abs(a2e.item() - ae.item()) < 1e-6 and abs(a2f.item() - af.item()) < 1e-6
    torch.testing.assert_close(eg.grad.cpu(), eo.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(fg.grad.cpu(), fo.grad, rtol=1e-3, atol=1e-5)
    for n, p in mg.named_parameters():
        if p.grad is None:
            continue
        torch.testing.assert_close(p.grad.cpu(), sd[n].grad, rtol=2e-3, atol=1e-5, msg=n)
    assert mg.logit_scale.grad is not None


def test_trainer_steps_reduce_loss_and_retrieve():
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    torch.manual_seed(0)
    tr = BridgeTrainer(eeg_channels=16, dropout=0.0, lr=1e-3).train()
    eeg, fmri = synthetic_pairs(16, 16, 256, (16, 16, 16))
    first = None
    for i in range(30):
        out = tr.train_step(eeg, fmri)
        if first is None:
            first = out["loss"].item()
    last = out["loss"].item()
    assert last < first * 0.7, (first, last)
    ev = tr.evaluate(eeg, fmri)
    assert ev["top1_e2f"].item() > 1.0 / 16
    assert float(tr.bucket.state[0]) == 30.0

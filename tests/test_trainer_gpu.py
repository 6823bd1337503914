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


def _oracle_step(tr, eeg, fmri, emulate):
    """the CPU oracle's forward + backward of one contrastive step on the trainer's current weights:
    (loss, ze, zf, {parameter name: gradient}) with names prefixed e. / f. / h. as in bench.cpu_baseline"""
    import contextlib
    from oracle.bf16_emulation import bf16_operands
    sd = {}
    for pre, m in (("e.", tr.eeg_encoder), ("f.", tr.fmri_encoder), ("h.", tr.head)):
        for k, v in m.state_dict().items():
            sd[pre + k] = v.detach().cpu().clone().requires_grad_(v.is_floating_point())
    with (bf16_operands() if emulate else contextlib.nullcontext()):
        fe = RF.erp_encoder(sd, eeg.cpu(), "e.", train=True)
        ff = RF.volume_encoder3d(sd, fmri.cpu(), "f.", train=True)
        ze, zf = RF.contrastive_head(sd, fe, ff, "h.bridge.")
        loss = RF.clip_loss(ze, zf, ze, zf, sd["h.logit_scale"].exp())[0]
        loss.backward()
    return loss.item(), ze.detach(), zf.detach(), {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}


def test_c2_shaped_train_step_vs_oracle():
    """BASELINE config #2 shapes (64 ch x 1024 EEG + 32^3 fMRI; B = 4 so that the CPU oracle finishes in
    seconds), dropout 0, the autograd-free tape run eagerly segment by segment - the step bench.py times:
    loss (1e-3), both L2-normalised embeddings (cos >= 1 - 1e-4 vs the fp32 oracle), every parameter gradient
    of the flat bucket: <= 6e-2 rel-L2 vs the oracle with bf16-rounded GEMM operands; the voxel encoder
    additionally <= 2e-1 vs pure fp32 (bf16 rounding flips 2x2x2 max-pool argmaxes)."""
    import torch.nn.functional as F
    from multimodal_eeg_fmri_amd import ops
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    ops.set_seed_epoch(None)
    torch.manual_seed(0)
    tr = BridgeTrainer(eeg_channels=64, dropout=0.0, mode="manual").train()
    eeg, fmri = synthetic_pairs(4, 64, 1024, (32, 32, 32), seed=4321)
    l32, ze32, zf32, g32 = _oracle_step(tr, eeg, fmri, emulate=False)
    l16, _, _, g16 = _oracle_step(tr, eeg, fmri, emulate=True)
    with torch.no_grad():
        z, saved = tr._seg_forward(eeg, fmri)
        dz = torch.empty_like(z)
        tr._seg_loss(z, tr._scal, dz)
        tr._seg_backward(saved, dz, tr._scal)
        ops.arena.end()
    torch.cuda.synchronize()
    N = tr.head.bridge.bridge_dim
    ze, zf = z[:, :N].cpu(), z[:, N:].cpu()
    cos_e = F.cosine_similarity(ze.double(), ze32.double(), dim=1).min().item()
    cos_f = F.cosine_similarity(zf.double(), zf32.double(), dim=1).min().item()
    assert cos_e >= 1 - 1e-4 and cos_f >= 1 - 1e-4, (cos_e, cos_f)
    assert abs(tr._scal[0].item() - l32) <= 1e-3 * max(1.0, abs(l32)), (tr._scal[0].item(), l32, l16)
    named = {}
    for pre, m in (("e.", tr.eeg_encoder), ("f.", tr.fmri_encoder), ("h.", tr.head)):
        named.update({pre + k: v for k, v in m.named_parameters()})
    worst16, worst32f = ("", 0.0), ("", 0.0)
    checked = 0
    for n, p in named.items():
        sink = getattr(p, "_mm_grad", None)
        if sink is None or n not in g16 or g16[n].norm() < 1e-5:
            continue
        got = sink.detach().cpu().view(g16[n].shape).double()
        e16 = ((got - g16[n].double()).norm() / g16[n].double().norm()).item()
        worst16 = max(worst16, (n, e16), key=lambda t: t[1])
        if n.startswith("f."):
            e32 = ((got - g32[n].double()).norm() / g32[n].double().norm()).item()
            worst32f = max(worst32f, (n, e32), key=lambda t: t[1])
        checked += 1
    assert checked >= 50, checked                      # conv biases in front of BatchNorm have ~0 gradient: skipped
    assert worst16[1] <= 6e-2, ("vs bf16-operand oracle", worst16)
    assert worst32f[1] <= 2e-1, ("voxel encoder vs fp32 oracle", worst32f)


def test_c2_graph_step_at_the_benchmarked_batch_vs_oracle():
    """the step bench.py times - B = 32, hipGraph replay, two streams - against the CPU oracle (the tests above
    run B = 4 on the eager tape).  Dropout 0 (the reference's RNG stream cannot be matched; the masked-oracle test covers
    p = 0.3 on the tape).  One replayed step from the initial weights: loss 1e-3, both L2-normalised embeddings cos >=
    1 - 1e-4 vs the fp32 oracle, every parameter gradient of the flat bucket (copied out by the step's ``grad_probe``
    node just before clip + AdamW clears it) <= 6e-2 rel-L2 vs the oracle with bf16-rounded GEMM operands, and the
    parameters after the step = torch.optim.AdamW + clip_grad_norm_(1.0) applied to the PROBED gradients (1e-6)."""
    import torch.nn.functional as F
    from multimodal_eeg_fmri_amd import ops
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    ops.set_seed_epoch(None)
    torch.manual_seed(0)
    tr = BridgeTrainer(eeg_channels=64, dropout=0.0, mode="graph").train()
    eeg, fmri = synthetic_pairs(32, 64, 1024, (32, 32, 32), seed=4323)
    l32, ze32, zf32, _ = _oracle_step(tr, eeg, fmri, emulate=False)
    _, _, _, g16 = _oracle_step(tr, eeg, fmri, emulate=True)
    p0 = tr.bucket.p.detach().clone()
    tr.grad_probe = torch.zeros_like(tr.bucket.g)
    out = tr.train_step(eeg, fmri)
    torch.cuda.synchronize()
    assert tr.capture_mode == "one graph" and len(tr._cap["graphs"]) == 1
    z = tr._cap["z"]
    N = tr.head.bridge.bridge_dim
    cos_e = F.cosine_similarity(z[:, :N].cpu().double(), ze32.double(), dim=1).min().item()
    cos_f = F.cosine_similarity(z[:, N:].cpu().double(), zf32.double(), dim=1).min().item()
    assert cos_e >= 1 - 1e-4 and cos_f >= 1 - 1e-4, (cos_e, cos_f)
    assert abs(out["loss"].item() - l32) <= 1e-3 * max(1.0, abs(l32)), (out["loss"].item(), l32)
    named = {}
    for pre, m in (("e.", tr.eeg_encoder), ("f.", tr.fmri_encoder), ("h.", tr.head)):
        named.update({pre + k: v for k, v in m.named_parameters()})
    base = tr.bucket.g.data_ptr()
    worst, checked = ("", 0.0), 0
    for n, p in named.items():
        sink = getattr(p, "_mm_grad", None)
        if sink is None or n not in g16 or g16[n].norm() < 1e-5:
            continue
        off = (sink.data_ptr() - base) // 4
        got = tr.grad_probe[off:off + p.numel()].cpu().view(g16[n].shape).double()
        worst = max(worst, (n, ((got - g16[n].double()).norm() / g16[n].double().norm()).item()), key=lambda t: t[1])
        checked += 1
    assert checked >= 50, checked
    assert worst[1] <= 6e-2, ("graph-replayed B = 32 step vs the bf16-operand oracle", worst)
    # the optimizer half of the same replay, on the probed gradients
    ref = p0.cpu().clone().requires_grad_(True)
    ref.grad = tr.grad_probe.cpu().clone()
    torch.nn.utils.clip_grad_norm_([ref], 1.0)
    opt = torch.optim.AdamW([ref], lr=tr.lr, weight_decay=tr.weight_decay, betas=tr.betas, eps=tr.eps)
    opt.step()
    torch.testing.assert_close(tr.bucket.p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("kind", ["stft", "power"])
def test_bridge_step_with_the_config5_eeg_branch_vs_oracle(kind):
    """BASELINE config #5 in the trainer: BridgeTrainer(eeg_encoder=MultiScaleSTFTPowerEncoder(...)) - raw EEG -> multi-scale
    STFT power (z-scored) -> EnhancedPowerEncoder - and a bare EnhancedPowerEncoder as the EEG branch, on the tape and in the
    hipGraph (dropout 0): one step against the CPU oracle (loss 2e-3; embeddings cos >= 1 - 1e-4 vs fp32; every parameter
    gradient of the flat bucket <= 7e-2 rel-L2 vs the oracle with bf16-rounded GEMM operands), and graph replay == eager
    tape bit for bit over four steps."""
    import torch.nn.functional as F
    from multimodal_eeg_fmri_amd import ops
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    from multimodal_eeg_fmri_amd.crossmodal_v4_enhancements import MultiScaleSTFTPowerEncoder
    from multimodal_eeg_fmri_amd.enhanced_models_v4 import EnhancedPowerEncoder
    from oracle.bf16_emulation import bf16_operands
    import contextlib
    C, T, vol, Bsz = 8, 256, (16, 16, 16), 8
    nffts, hop = (16, 32), 8

    def make(mode):
        ops.set_seed_epoch(None)
        torch.manual_seed(0)
        enc = MultiScaleSTFTPowerEncoder(C, nffts, hop, 128, 2, 4, 0.0) if kind == "stft" else EnhancedPowerEncoder(C, 128, 2, 4, 0.0)
        return BridgeTrainer(eeg_channels=C, dropout=0.0, lr=1e-3, mode=mode, eeg_encoder=enc).train()

    eeg, fmri = synthetic_pairs(Bsz, C, T, vol, seed=4400)
    tr = make("manual")
    assert tr._eeg_kind == kind

    def oracle(emulate):
        sd = {}
        for pre, m in (("e.", tr.eeg_encoder), ("f.", tr.fmri_encoder), ("h.", tr.head)):
            for k, v in m.state_dict().items():
                sd[pre + k] = v.detach().cpu().clone().requires_grad_(v.is_floating_point())
        with (bf16_operands() if emulate else contextlib.nullcontext()):
            if kind == "stft":
                fe = RF.stft_power_encoder(sd, eeg.cpu(), nffts, hop, "e.encoder.", train=True)
            else:
                fe = RF.power_encoder(sd, eeg.cpu(), "e.", train=True)
            ff = RF.volume_encoder3d(sd, fmri.cpu(), "f.", train=True)
            ze, zf = RF.contrastive_head(sd, fe, ff, "h.bridge.")
            loss = RF.clip_loss(ze, zf, ze, zf, sd["h.logit_scale"].exp())[0]
            loss.backward()
        return loss.item(), ze.detach(), zf.detach(), {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    l32, ze32, zf32, _ = oracle(False)
    _, _, _, g16 = oracle(True)
    with torch.no_grad():
        z, saved = tr._seg_forward(eeg, fmri)
        dz = torch.empty_like(z)
        tr._seg_loss(z, tr._scal, dz)
        tr._seg_backward(saved, dz, tr._scal)
        ops.arena.end()
    torch.cuda.synchronize()
    N = tr.head.bridge.bridge_dim
    cos_e = F.cosine_similarity(z[:, :N].cpu().double(), ze32.double(), dim=1).min().item()
    cos_f = F.cosine_similarity(z[:, N:].cpu().double(), zf32.double(), dim=1).min().item()
    assert cos_e >= 1 - 1e-4 and cos_f >= 1 - 1e-4, (cos_e, cos_f)
    assert abs(tr._scal[0].item() - l32) <= 2e-3 * max(1.0, abs(l32)), (tr._scal[0].item(), l32)
    named = {}
    for pre, m in (("e.", tr.eeg_encoder), ("f.", tr.fmri_encoder), ("h.", tr.head)):
        named.update({pre + k: v for k, v in m.named_parameters()})
    worst, checked = ("", 0.0), 0
    for n, p in named.items():
        sink = getattr(p, "_mm_grad", None)
        if sink is None or n not in g16 or g16[n].norm() < 1e-5:
            continue
        got = sink.detach().cpu().view(g16[n].shape).double()
        worst = max(worst, (n, ((got - g16[n].double()).norm() / g16[n].double().norm()).item()), key=lambda t: t[1])
        checked += 1
    assert checked >= 45, checked
    assert worst[1] <= 7e-2, ("vs the bf16-operand oracle", worst)
    # graph replay == eager tape, bit for bit
    res = []
    for mode in ("manual", "graph"):
        t2 = make(mode)
        losses = [t2.train_step(eeg, fmri)["loss"].clone() for _ in range(4)]
        torch.cuda.synchronize()
        res.append((torch.stack(losses), t2.bucket.p.detach().clone()))
        ops.set_seed_epoch(None)
    assert torch.isfinite(res[0][0]).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_projection_heads_with_dropout_match_masked_oracle():
    """both projection heads' nn.Dropout (bridge_utils.py:34-45) at p = 0.3 in ONE launch each way
    (mm_proj_heads_fwd / _bwd): embeddings, loss and every gradient against the CPU oracle evaluated with the same
    counter-hash masks; fp32 kernels -> 1e-4 / 2e-3."""
    from multimodal_eeg_fmri_amd import ops
    from oracle.dropout_replica import contrastive_head_with_masks
    p = 0.3
    m = build(B.EEGfMRIContrastiveBridge, 43, dropout=p).train()
    eeg, fmri = seeded_randn(143, 16, 128), seeded_randn(144, 16, 64)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    seeds = []
    real = ops._next_seed

    def logged():
        v = real()
        seeds.append(v)
        return v
    ops.set_seed_epoch(None)
    ops.set_dropout_seed(99)
    ops._next_seed = logged
    try:
        mg = m.cuda()
        eg, fg = eeg.cuda().requires_grad_(True), fmri.cuda().requires_grad_(True)
        l2, _, _ = mg(eg, fg)
        l2.backward()
    finally:
        ops._next_seed = real
    assert len(seeds) == 2, seeds
    eo, fo = eeg.clone().requires_grad_(True), fmri.clone().requires_grad_(True)
    ze, zf = contrastive_head_with_masks(sd, eo, fo, seeds, p, "bridge.")
    loss = RF.clip_loss(ze, zf, ze, zf, sd["logit_scale"].exp())[0]
    loss.backward()
    assert 0.2 < (ze == 0).float().mean().item() < 0.4
    assert abs(l2.item() - loss.item()) < 1e-4
    torch.testing.assert_close(eg.grad.cpu(), eo.grad, rtol=2e-3, atol=1e-5)
    torch.testing.assert_close(fg.grad.cpu(), fo.grad, rtol=2e-3, atol=1e-5)
    for n, q in mg.named_parameters():
        if q.grad is not None:
            torch.testing.assert_close(q.grad.cpu(), sd[n].grad, rtol=2e-3, atol=1e-5, msg=n)


def test_c2_shaped_train_step_with_dropout_vs_masked_oracle():
    """the configuration bench.py times - BASELINE config #2 shapes, dropout 0.3 at all 19 sites (B = 4 so that the CPU
    oracle finishes in seconds) - run as the eager tape, against the CPU oracle evaluated with the SAME keep-masks
    (oracle/dropout_replica.py: bridge_step_with_masks): loss 2e-3, both embeddings cos >= 1 - 1e-4, every parameter
    gradient of the flat bucket <= 7e-2 rel-L2 vs the oracle with bf16-rounded GEMM operands."""
    import torch.nn.functional as F
    from multimodal_eeg_fmri_amd import ops
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    from oracle.bf16_emulation import bf16_operands
    from oracle.dropout_replica import bridge_step_with_masks
    p = 0.3
    torch.manual_seed(0)
    tr = BridgeTrainer(eeg_channels=64, dropout=p, mode="manual").train()
    eeg, fmri = synthetic_pairs(4, 64, 1024, (32, 32, 32), seed=4322)
    sd = {}
    for pre, m in (("e.", tr.eeg_encoder), ("f.", tr.fmri_encoder), ("h.", tr.head)):
        for k, v in m.state_dict().items():
            sd[pre + k] = v.detach().cpu().clone().requires_grad_(v.is_floating_point())
    seeds = []
    real = ops._next_seed

    def logged():
        v = real()
        seeds.append(v)
        return v
    ops.set_seed_epoch(None)
    ops.set_dropout_seed(31337)
    ops._next_seed = logged
    try:
        with torch.no_grad():
            z, saved = tr._seg_forward(eeg, fmri)
            dz = torch.empty_like(z)
            tr._seg_loss(z, tr._scal, dz)
            tr._seg_backward(saved, dz, tr._scal)
            ops.arena.end()
    finally:
        ops._next_seed = real
    torch.cuda.synchronize()
    assert len(seeds) == 19, seeds
    with bf16_operands():
        loss, ze16, zf16 = bridge_step_with_masks(sd, eeg.cpu(), fmri.cpu(), seeds, p)
        loss.backward()
    N = tr.head.bridge.bridge_dim
    ze, zf = z[:, :N].cpu(), z[:, N:].cpu()
    assert torch.equal(ze == 0, ze16 == 0) and torch.equal(zf == 0, zf16 == 0)          # the heads' masks are the replica's
    cos_e = F.cosine_similarity(ze.double(), ze16.detach().double(), dim=1).min().item()
    cos_f = F.cosine_similarity(zf.double(), zf16.detach().double(), dim=1).min().item()
    assert cos_e >= 1 - 1e-4 and cos_f >= 1 - 1e-4, (cos_e, cos_f)
    assert abs(tr._scal[0].item() - loss.item()) <= 2e-3 * max(1.0, abs(loss.item())), (tr._scal[0].item(), loss.item())
    named = {}
    for pre, m in (("e.", tr.eeg_encoder), ("f.", tr.fmri_encoder), ("h.", tr.head)):
        named.update({pre + k: v for k, v in m.named_parameters()})
    worst, checked = ("", 0.0), 0
    for n, q in named.items():
        sink = getattr(q, "_mm_grad", None)
        want = sd[n].grad if n in sd else None
        if sink is None or want is None or want.norm() < 1e-5:
            continue
        got = sink.detach().cpu().view(want.shape).double()
        worst = max(worst, (n, ((got - want.double()).norm() / want.double().norm()).item()), key=lambda t: t[1])
        checked += 1
    assert checked >= 50, checked
    assert worst[1] <= 7e-2, ("vs the masked bf16-operand oracle", worst)


def test_step_results_are_owned_by_the_trainer():
    """the returned loss / top-1 tensors must not alias scratch another trainer rewrites"""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    torch.manual_seed(0)
    a = BridgeTrainer(eeg_channels=16, dropout=0.0, lr=1e-3, mode="graph").train()
    b = BridgeTrainer(eeg_channels=16, dropout=0.0, lr=1e-2, mode="graph").train()
    eeg, fmri = synthetic_pairs(8, 16, 256, (16, 16, 16))
    for _ in range(3):
        out_a = a.train_step(eeg, fmri)
    la = out_a["loss"].item()
    for _ in range(5):
        b.train_step(eeg, fmri)
    torch.cuda.synchronize()
    assert out_a["loss"].item() == la
    # a loader that fills only ONE of the static input buffers in place: the other must still be refreshed
    bufs = a.input_buffers()
    eeg2, fmri2 = synthetic_pairs(8, 16, 256, (16, 16, 16), seed=99)
    bufs[0].copy_(eeg2)
    l_new = a.train_step(bufs[0], fmri2)["loss"].item()
    assert torch.equal(a.input_buffers()[1], fmri2)
    assert l_new != la


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


@pytest.mark.parametrize("mode", ["autograd", "manual", "graph"])
def test_trainer_modes_agree_on_first_steps(mode):
    """the autograd surface, the autograd-free tape and the hipGraph replay run the
    same kernels: with dropout 0 their losses over the first steps agree."""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    from multimodal_eeg_fmri_amd import ops
    ops.set_seed_epoch(None)
    eeg, fmri = synthetic_pairs(8, 16, 256, (16, 16, 16))

    def run(m):
        torch.manual_seed(0)
        tr = BridgeTrainer(eeg_channels=16, dropout=0.0, lr=1e-3, mode=m).train()
        losses = []
        for _ in range(6):
            losses.append(tr.train_step(eeg, fmri)["loss"].item())
        ops.set_seed_epoch(None)
        return losses
    ref = run("autograd")
    got = run(mode)
    for a, b in zip(ref, got):
        assert abs(a - b) <= 2e-2 * max(1.0, abs(a)), (ref, got)


@pytest.mark.parametrize("mode,shape", [("graph", (32, 64, 1024, (32, 32, 32))), ("manual", (8, 16, 256, (16, 16, 16)))])
def test_training_is_bit_reproducible(mode, shape):
    """no floating-point atomics anywhere in the step - weight gradients go through per-workgroup slots
    summed in order, per-channel sums through 64-bit fixed-point accumulators (integer adds commute), everything else
    has one writer.  Two trainers built from the same seeds and fed the same batches (the C2 bench step, dropout on,
    hipGraph replay with its two streams; and the eager tape) hold BIT-IDENTICAL parameters, optimizer state and
    losses after 10 steps."""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    from multimodal_eeg_fmri_amd import ops
    Bsz, C, T, vol = shape
    batches = [synthetic_pairs(Bsz, C, T, vol, seed=100 + i) for i in range(3)]

    def run():
        ops.set_seed_epoch(None)
        ops.set_dropout_seed(1234)
        torch.manual_seed(0)
        tr = BridgeTrainer(eeg_channels=C, dropout=0.1, lr=1e-3, mode=mode).train()
        losses = []
        for i in range(10):
            eeg, fmri = batches[i % 3]
            losses.append(tr.train_step(eeg, fmri)["loss"].clone())
        torch.cuda.synchronize()
        params = [p.detach().clone() for m in (tr.eeg_encoder, tr.fmri_encoder, tr.head) for p in m.parameters()]
        bufs = [b.detach().clone() for m in (tr.eeg_encoder, tr.fmri_encoder, tr.head) for b in m.buffers()]
        state = tr.bucket.state.detach().clone()
        ops.set_seed_epoch(None)
        return torch.stack(losses), params, bufs, state

    l1, p1, b1, s1 = run()
    l2, p2, b2, s2 = run()
    assert torch.isfinite(l1).all()
    assert torch.equal(l1, l2), (l1 - l2).abs().max().item()
    assert len(p1) == len(p2) and all(torch.equal(a, b) for a, b in zip(p1, p2)), \
        max((a - b).abs().max().item() for a, b in zip(p1, p2))
    assert all(torch.equal(a, b) for a, b in zip(b1, b2))          # BatchNorm running statistics
    assert torch.equal(s1, s2)


def test_fused_launches_train_exactly_as_their_separate_forms(monkeypatch):
    """Round 3's launch fusions are re-arrangements, not approximations: with the second-GEMM forms (out-projection data
    gradient, next block's QKV projection, first FFN Linear), the one-row-per-sample head gradients and either hand-over
    of the conv weight gradients switched, six steps at the C2 shape (dropout on, hipGraph) give BIT-IDENTICAL losses and
    parameters; the BatchNorm-reduce epilogues regroup fp32 partial sums, so they are held to rounding level."""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    from multimodal_eeg_fmri_amd import autograd, ops
    batches = [synthetic_pairs(32, 64, 1024, (32, 32, 32), seed=300 + i) for i in range(2)]

    def run(env=None, **knobs):
        if env:
            monkeypatch.setenv(*env)
        for k, v in knobs.items():
            mod, name = (ops, k[4:]) if k.startswith("ops_") else (autograd, k)
            monkeypatch.setattr(mod, name, v)
        ops.set_seed_epoch(None)
        ops.set_dropout_seed(4321)
        torch.manual_seed(0)
        tr = BridgeTrainer(eeg_channels=64, dropout=0.2, lr=1e-3).train()
        losses = [tr.train_step(*batches[i % 2])["loss"].clone() for i in range(6)]
        torch.cuda.synchronize()
        params = torch.cat([p.detach().flatten() for m in (tr.eeg_encoder, tr.fmri_encoder, tr.head) for p in m.parameters()]).clone()
        ops.set_seed_epoch(None)
        monkeypatch.undo()
        return torch.stack(losses), params
    l0, p0 = run()
    assert torch.isfinite(l0).all()
    for knobs in (dict(_NO_GEMM2=True), dict(ops__NO_QKV_FUSE=True), dict(_NO_BCAST=True), dict(ops__NO_FFN1_FUSE=False),
                  dict(_NO_GEMM2=True, _NO_BCAST=True, ops__NO_QKV_FUSE=True)):
        l1, p1 = run(**knobs)
        assert torch.equal(l0, l1) and torch.equal(p0, p1), knobs
    l2, p2 = run(env=("MM_CONV_WGRADS_HANDED", "2"))
    assert torch.equal(l0, l2) and torch.equal(p0, p2)
    l3, p3 = run(_NO_BNRED=True)
    torch.testing.assert_close(l3, l0, rtol=2e-4, atol=2e-4)
    assert ((p3 - p0).norm() / p0.norm()).item() < 2e-3


def test_config4_volume_step_takes_the_fmri_first_schedule_and_matches_the_other(monkeypatch):
    """config #4's 64 x 64 x 48 volumes make the voxel branch the longer stream: the trainer picks the fMRI-first
    schedule by itself (`_fmri_is_longer`), and four graph steps end bit-identical to the EEG-first / hand-over schedule
    forced by MM_FMRI_LONGER=0 (what config #2 runs, which the oracle tests pin).  Dropout off: the masks' seeds are
    drawn in issue order, so the two schedules would draw different (equally valid) masks."""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    from multimodal_eeg_fmri_amd import ops
    batches = [synthetic_pairs(4, 16, 256, (64, 64, 48), seed=500 + i) for i in range(2)]

    def run(env):
        if env is not None:
            monkeypatch.setenv("MM_FMRI_LONGER", env)
        ops.set_seed_epoch(None)
        ops.set_dropout_seed(99)
        torch.manual_seed(0)
        tr = BridgeTrainer(eeg_channels=16, dropout=0.0, lr=1e-3).train()
        losses = [tr.train_step(*batches[i % 2])["loss"].clone() for i in range(4)]
        torch.cuda.synchronize()
        params = tr.bucket.p.detach().clone()
        ops.set_seed_epoch(None)
        monkeypatch.undo()
        return torch.stack(losses), params, tr._fmri_longer
    la, pa, longer = run(None)
    assert longer and torch.isfinite(la).all()
    lb, pb, longer_b = run("0")
    assert not longer_b
    assert torch.equal(la, lb) and torch.equal(pa, pb)
    assert not BridgeTrainer._fmri_is_longer(torch.empty(2, 1, 32, 32, 32))


def test_packed_host_batch_step_is_bit_identical_to_the_device_side_pack():
    """the host-fed path: BridgeTrainer.pack_host_batch (CPU: EEG epochs into the first convolution's bf16 channels-last
    operand, round-to-nearest-even, + fp32 volumes, ONE flat buffer) followed by train_step_packed (one D2D copy into the
    step's static inputs + replay) trains BIT-identically to train_step on the fp32 device tensors (mm_stage_inputs packs on
    the device), dropout on; a buffer of the wrong size is refused.  So does the loop a loader drives (`HostFeeder`: H2D
    copies one step ahead on a copy stream, ordered from the host)."""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    from multimodal_eeg_fmri_amd import ops
    batches = [synthetic_pairs(8, 16, 256, (16, 16, 16), seed=700 + i) for i in range(3)]

    def run(packed):
        ops.set_seed_epoch(None)
        ops.set_dropout_seed(77)
        torch.manual_seed(0)
        tr = BridgeTrainer(eeg_channels=16, dropout=0.2, lr=1e-3).train()
        losses = [tr.train_step(*batches[0])["loss"].clone()]          # captures; fixes the shapes
        if packed == "feeder":                                          # the loop a loader drives: copies one step ahead
            feeder = tr.host_feeder()
            hosts = [tr.pack_host_batch(*batches[i % 3]) for i in range(1, 6)]
            feeder.upload(hosts[0])
            for i in range(5):
                if i + 1 < 5:
                    feeder.upload(hosts[i + 1])
                losses.append(feeder.step()["loss"].clone())
            with pytest.raises(RuntimeError, match="nothing uploaded"):
                feeder.step()
            with pytest.raises(ValueError):
                feeder.upload(torch.zeros(10, dtype=torch.uint8))
        for i in range(1, 6 if packed != "feeder" else 1):
            e, f = batches[i % 3]
            if packed:
                host = tr.pack_host_batch(e, f)
                assert host.is_pinned() and host.dtype == torch.uint8 and host.numel() == 8 * 256 * 16 * 2 + 8 * 16 ** 3 * 4
                losses.append(tr.train_step_packed(host.cuda(non_blocking=True))["loss"].clone())
            else:
                losses.append(tr.train_step(e, f)["loss"].clone())
        torch.cuda.synchronize()
        p = tr.bucket.p.detach().clone()
        ops.set_seed_epoch(None)
        return torch.stack(losses), p, tr
    l0, p0, _ = run(False)
    l1, p1, tr = run(True)
    assert torch.isfinite(l0).all() and torch.equal(l0, l1) and torch.equal(p0, p1)
    l2, p2, _ = run("feeder")
    assert torch.equal(l0, l2) and torch.equal(p0, p2)
    with pytest.raises(ValueError):
        tr.train_step_packed(torch.zeros(100, dtype=torch.uint8, device="cuda"))


def test_graph_mode_draws_new_dropout_masks_each_replay():
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
    from multimodal_eeg_fmri_amd import ops
    torch.manual_seed(0)
    tr = BridgeTrainer(eeg_channels=16, dropout=0.5, lr=0.0, weight_decay=0.0, mode="graph").train()
    eeg, fmri = synthetic_pairs(8, 16, 256, (16, 16, 16))
    losses = [tr.train_step(eeg, fmri)["loss"].item() for _ in range(4)]
    ops.set_seed_epoch(None)
    assert len({round(l, 5) for l in losses}) > 1, losses      # lr = 0: only the masks change


def test_two_rank_data_parallel_step_on_one_gpu():
    """a-X5: two ranks share this GPU, exchange steps over gloo (RCCL refuses two
    ranks per device).  Same code the N>1 bench runs: 3 hipGraph segments + 2
    collectives.  Ranks must end bit-identical; graph replay must equal the eager tape."""
    from tools import dp_rehearsal as mod
    r = mod.run(2)
    assert r["same_params_across_ranks"], r
    # the step is bit-reproducible (no floating-point atomics): the segmented graph replay and the eager tape, six steps
    # each from the same state, end on IDENTICAL parameters and report identical losses
    assert r["graph_vs_manual_rel"] == 0.0, r
    assert r["losses"]["graph"] == r["losses"]["manual"], r
    assert r["losses"]["graph"][-1] < r["losses"]["graph"][0], r


def test_segmented_step_through_rccl_with_one_rank():
    """a-X5: the N > 1 code path (3 hipGraph segments, the two collectives between them) through
    the real backend ("nccl" == RCCL) at world size 1, in a child process: RCCL initialises, every
    collective call of the multi-GPU bench is accepted, and the result equals the one-graph step."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dp_rehearsal.py"), "rccl1"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_share_the_gpu_over_gloo(launcher):
    """bench.py at N = 2 both ways the driver may start it - `python bench.py --gpus 2` with no launcher (it then starts
    its own ranks before touching the GPU and relays rank 0's line) and under torch.distributed.run - except that both
    ranks share this GPU and exchange over gloo (MM_DIST_BACKEND): the N > 1 timing / reduction / JSON path must run,
    report whole-job throughput and carry the N > 1 fields."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MM_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29517"] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["scaling"] == "weak"
    assert line["value"] > 0 and abs(line["value"] - 64 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert line["rccl_ranks"] == 0 and line["capture_mode"] == "3 segments + 2 eager collectives"      # gloo rehearsal
    assert len(line["gradient_bucket_groups"]) == 4 and len(line["collectives_us"]) == 5
    assert abs(sum(g["MB"] for g in line["gradient_bucket_groups"]) - 3.4) < 0.2

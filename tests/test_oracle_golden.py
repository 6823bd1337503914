"""CPU: the functional oracle (oracle/ref_functional.py) reproduces every golden
vector generated from the reference (oracle/make_goldens.py), with weights
rebuilt from seeds on the PRODUCT classes (checksummed against the reference's).
"""
import numpy as np
import pytest
import torch

from oracle import ref_functional as RF
from oracle.fixtures import build, checksum, seeded_randn

import multimodal_eeg_fmri_amd.bridge_utils as B
import multimodal_eeg_fmri_amd.crossmodal_v4_enhancements as C
import multimodal_eeg_fmri_amd.enhanced_models_v4 as E
import multimodal_eeg_fmri_amd.fmri_utils as Fm

TOL = 2e-6


def _chk(model, fx, key="cks"):
    np.testing.assert_allclose(checksum(model), fx[key], rtol=1e-6, atol=1e-6,
                               err_msg="seeded weights differ from the reference's")


def _eq(a, b, tol=TOL):
    b = torch.as_tensor(b)
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_a3_erp_encoder(golden, tag):
    fx = golden(f"a3_erp_{tag}.npz")
    Bn, Cn, T = (int(v) for v in fx["shape"])
    m = build(E.EnhancedERPEncoder, int(fx["seed"]), Cn, 128, 2, 4, 0.3).eval()
    _chk(m, fx)
    x = seeded_randn(int(fx["x_seed"]), Bn, Cn, T)
    st = {}
    with torch.no_grad():
        y = RF.erp_encoder(m.state_dict(), x, stages=st)
    _eq(y, fx["out"])
    s = int(fx["stride"])
    for k, v in st.items():
        sub = v[:, :, ::s] if k.startswith("conv") else v[:, ::s, :]
        _eq(sub, fx["stage_" + k])


def test_a4_power_encoder(golden):
    fx = golden("a4_power.npz")
    m = build(E.EnhancedPowerEncoder, int(fx["seed"]), 64, 128, 2, 4, 0.3).eval()
    _chk(m, fx)
    x = seeded_randn(int(fx["x_seed"]), *[int(v) for v in fx["shape"]])
    with torch.no_grad():
        _eq(RF.power_encoder(m.state_dict(), x), fx["out"])


def test_a7_trimodal_lite(golden):
    fx = golden("a7_lite.npz")
    m = build(C.EnhancedTriModalFusionNetV4Lite, int(fx["seed"]), 8, 8, 459).eval()
    _chk(m, fx)
    s = [int(v) for v in fx["x_seeds"]]
    erp, pw, conn = seeded_randn(s[0], 8, 8, 256), seeded_randn(s[1], 8, 8, 256), seeded_randn(s[2], 8, 459)
    sd = m.state_dict()
    with torch.no_grad():
        logits, fused, _ = RF.trimodal_lite(sd, erp, pw, conn)
        _eq(logits, fx["logits"])
        _eq(fused, fx["fused"])
        _eq(RF.lite_erp_encoder(sd, erp, "erp_encoder."), fx["erp_feat"])
        _eq(RF.lite_power_encoder(sd, pw, "pw_encoder."), fx["pw_feat"])
        _eq(RF.conn_encoder(sd, conn, "conn_encoder."), fx["conn_feat"])


def test_a9_fmri_fusion(golden):
    fx = golden("a9_fmri.npz")
    m = build(Fm.fMRIFusionNet, int(fx["seed"]), 100, 200).eval()
    _chk(m, fx)
    s = [int(v) for v in fx["x_seeds"]]
    with torch.no_grad():
        out, fused = RF.fmri_fusion_net(m.state_dict(), seeded_randn(s[0], 8, 100), seeded_randn(s[1], 8, 200))
    _eq(out, fx["out"])
    _eq(fused, fx["fused"])


def test_a11_bridge(golden):
    fx = golden("a11_bridge.npz")
    m = build(B.EEGfMRIBridgeFusionNet, int(fx["seed"])).eval()
    _chk(m, fx)
    s = [int(v) for v in fx["x_seeds"]]
    eeg, fmri = seeded_randn(s[0], 8, 128), seeded_randn(s[1], 8, 64)
    sd = m.state_dict()
    with torch.no_grad():
        logits, fused, fw, aw = RF.bridge_net(sd, eeg, fmri)
        _eq(RF.projection_head(sd, "eeg_proj.", eeg), fx["eeg_proj"])
        _eq(RF.projection_head(sd, "fmri_proj.", fmri), fx["fmri_proj"])
    _eq(logits, fx["logits"]); _eq(fused, fx["fused"]); _eq(fw, fx["fusion_w"]); _eq(aw, fx["attn_w"])
    g = m.get_fusion_weights()
    np.testing.assert_allclose([g["eeg_weight"], g["fmri_weight"], g["temperature"]], fx["gfw"], atol=1e-7)


@pytest.mark.parametrize("M", [2, 3])
def test_a5_learned_fusion(golden, M):
    fx = golden("a5_fusion.npz")
    m = build(E.LearnedFusionModule, 16 + M, M, 128, perturb=False).eval()
    with torch.no_grad():
        m.fusion_logits.copy_(torch.linspace(0.5, 1.5, M))
        m.temperature.fill_(0.7)
    _chk(m, fx, f"cks{M}")
    feats = [seeded_randn(110 + i, 8, 128) for i in range(M)]
    with torch.no_grad():
        f, w = RF.learned_fusion(m.state_dict(), feats)
    _eq(f, fx[f"fused{M}"]); _eq(w, fx[f"w{M}"])


def test_a3_train_mode_gradients(golden):
    fx = golden("a3_erp_train_grads.npz")
    m = build(E.EnhancedERPEncoder, int(fx["seed"]), 8, 128, 2, 4, 0.0).train()
    _chk(m, fx)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    x = seeded_randn(int(fx["x_seed"]), 8, 8, 256).requires_grad_(True)
    y = RF.erp_encoder(sd, x, train=True)
    y.backward(seeded_randn(int(fx["gy_seed"]), 8, 128))
    _eq(y.detach(), fx["out"], 5e-6)
    _eq(x.grad, fx["dx"], 2e-5)
    for n, gn in zip(fx["gnames"], fx["gnorms"]):
        g = sd[str(n)].grad
        assert abs(g.double().norm().item() - gn) <= 5e-5 * max(1.0, gn), n
        _eq(g.flatten()[:64], fx["ghead::" + str(n)], 5e-5)


def test_a11_train_mode_gradients(golden):
    fx = golden("a11_bridge_train_grads.npz")
    m = build(B.EEGfMRIBridgeFusionNet, int(fx["seed"]), dropout=0.0).train()
    _chk(m, fx)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    s = [int(v) for v in fx["x_seeds"]]
    eeg = seeded_randn(s[0], 8, 128).requires_grad_(True)
    fmri = seeded_randn(s[1], 8, 64).requires_grad_(True)
    logits = RF.bridge_net(sd, eeg, fmri)[0]
    loss = torch.nn.functional.cross_entropy(logits, torch.as_tensor(fx["target"]),
                                             weight=torch.as_tensor(fx["class_w"]))
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) < 1e-6
    _eq(eeg.grad, fx["d_eeg"], 1e-5); _eq(fmri.grad, fx["d_fmri"], 1e-5)
    for n, gn in zip(fx["gnames"], fx["gnorms"]):
        g = sd[str(n)].grad
        assert abs(g.double().norm().item() - gn) <= 2e-5 * max(1.0, gn), n


def test_a8_training_utilities(golden):
    fx = golden("a8_train_utils.npz")
    logits, tgt = torch.as_tensor(fx["logits"]), torch.as_tensor(fx["target"])
    assert abs(RF.label_smoothing_ce(logits, tgt, 0.1).item() - float(fx["ls_loss"])) < 1e-6
    lrs = [RF.cosine_warmup_lr(e, 5e-5, 3, 50) for e in range(1, 51)]
    np.testing.assert_allclose(lrs, fx["lrs"], rtol=1e-12)
    # host-side product classes (pure Python) against the same pins
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=5e-5)
    sch = C.CosineAnnealingWarmup(opt, warmup_epochs=3, total_epochs=50)
    np.testing.assert_allclose([sch.step() for _ in range(50)], fx["lrs"], rtol=1e-12)
    es = C.EarlyStopping(patience=3, mode="max")
    assert [bool(es(float(s))) for s in fx["es_scores"]] == [bool(v) for v in fx["es_stops"]]


def test_extension_restatements_match_torch_leaf_ops():
    """a-X1/a-X2/a-X3 have no reference counterpart ("parity unpinned"): pin the
    restatement to torch's own leaf modules instead."""
    m = build(Fm.fMRIVolumeEncoder3D, 31).eval()
    x = seeded_randn(131, 2, 1, 16, 16, 16)
    with torch.no_grad():
        want = m.output_proj(m.conv_layers(x))      # CPU leaf modules, checker only
        got = RF.volume_encoder3d(m.state_dict(), x)
    _eq(got, want)
    ze = RF.l2_normalize(seeded_randn(132, 6, 128)); zf = RF.l2_normalize(seeded_randn(133, 6, 128))
    loss, acc_e, acc_f, s = RF.clip_loss(ze, zf, ze, zf, torch.tensor(14.0))
    tgt = torch.arange(6)
    want = 0.5 * (torch.nn.functional.cross_entropy(14.0 * ze @ zf.t(), tgt)
                  + torch.nn.functional.cross_entropy(14.0 * zf @ ze.t(), tgt))
    assert abs(loss.item() - want.item()) < 1e-6
    p = RF.stft_power(seeded_randn(134, 2, 3, 256), 64, 32)
    assert p.shape == (2, 3 * 33, 9) and bool((p >= 0).all())


@pytest.mark.parametrize("tag,name,n_in", [("trimodal_v4", "EnhancedTriModalFusionNetV4", 3),
                                            ("smart_v4", "EnhancedSmartFusionNetV4", 2)])
def test_f1_full_v4_classifiers(golden, tag, name, n_in):
    """the oracle's restatement of the full V4 classifiers against the reference's own
    outputs (eval) and gradient norms (train, dropout 0)."""
    fx = golden(f"f1_{tag}.npz")
    args = (8, 8, 36) if n_in == 3 else (8, 8)
    m = build(getattr(C, name), int(fx["seed"]), *args).eval()
    _chk(m, fx)
    s = [int(v) for v in fx["x_seeds"]]
    xs = [seeded_randn(s[0], 4, 8, 256), seeded_randn(s[1], 4, 8, 256)] + ([seeded_randn(s[2], 4, 36)] if n_in == 3 else [])
    fn = RF.trimodal_v4 if n_in == 3 else RF.smart_fusion_v4
    with torch.no_grad():
        logits, weights, fused = fn(m.state_dict(), *xs)
    _eq(logits, fx["logits"], 5e-6); _eq(weights, fx["weights"], 5e-6); _eq(fused, fx["fused"], 5e-6)
    mt = build(getattr(C, name), int(fx["train_seed"]), *args, dropout=0.0).train()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in mt.state_dict().items()}
    fn(sd, *xs, train=True)[0].backward(seeded_randn(int(fx["gy_seed"]), 4, 2))
    for n, gn in zip((str(v) for v in fx["t_gnames"]), fx["t_gnorms"]):
        if gn < 1e-4:          # biases in front of a train-mode BatchNorm: true gradient 0, rounding noise only
            continue
        got = sd[n].grad.double().norm().item()
        assert abs(got - gn) <= 1e-4 * gn, (n, got, gn)


def test_a1_a2_standalone_pe_and_masked_block_vs_reference_golden(golden):
    """round-2 fixture (oracle/make_goldens_r2.py): PositionalEncoding.forward in both layout branches
    and TemporalTransformerBlock.forward(x, mask) for mask = None / boolean causal / additive float."""
    import multimodal_eeg_fmri_amd.enhanced_models_v4 as E
    from oracle.make_goldens_r2 import masks
    fx = golden("a1a2_standalone.npz")
    pe = E.PositionalEncoding(128, dropout=0.1).pe
    s1, s2 = (int(v) for v in fx["pe_x_seeds"])
    xb, xs = seeded_randn(s1, 2, 96, 128), seeded_randn(s2, 40, 1, 128)
    torch.testing.assert_close(RF.positional_encoding({"pe": pe}, "", xb), torch.as_tensor(fx["pe_out_bf"]), rtol=0, atol=1e-6)
    torch.testing.assert_close(RF.positional_encoding({"pe": pe}, "", xs), torch.as_tensor(fx["pe_out_sf"]), rtol=0, atol=1e-6)
    m = build(E.TemporalTransformerBlock, int(fx["blk_seed"]), 128, 4, 512, 0.1).eval()
    np.testing.assert_allclose(checksum(m), fx["blk_cks"], rtol=1e-6, atol=1e-6)
    x = seeded_randn(int(fx["blk_x_seed"]), 2, 96, 128)
    mk = masks(96)
    torch.testing.assert_close(mk["float"], torch.as_tensor(fx["mask_float"]))
    for tag, msk in mk.items():
        with torch.no_grad():
            y = RF.transformer_block(m.state_dict(), "", x, 4, mask=msk)
        torch.testing.assert_close(y, torch.as_tensor(fx[f"blk_out_{tag}"]), rtol=1e-5, atol=1e-5)


def test_a3_train_grads_at_c2_shape_vs_reference_golden(golden):
    """the oracle's train-mode forward and every gradient at the shape bench.py times (64 ch x 1024, B = 2)"""
    import multimodal_eeg_fmri_amd.enhanced_models_v4 as E
    fx = golden("a3_erp_train_grads_c2.npz")
    B, C, T = (int(v) for v in fx["shape"])
    m = build(E.EnhancedERPEncoder, int(fx["seed"]), C, 128, 2, 4, 0.0).train()
    np.testing.assert_allclose(checksum(m), fx["cks"], rtol=1e-6, atol=1e-6)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    x = seeded_randn(int(fx["x_seed"]), B, C, T).requires_grad_(True)
    y = RF.erp_encoder(sd, x, train=True)
    y.backward(seeded_randn(int(fx["gy_seed"]), B, 128))
    torch.testing.assert_close(y.detach(), torch.as_tensor(fx["out"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(x.grad[:, :, ::8], torch.as_tensor(fx["dx_t8"]), rtol=1e-3, atol=1e-5)
    for n, gn in zip((str(n) for n in fx["gnames"]), fx["gnorms"]):
        if gn < 1e-4:
            continue
        assert abs(sd[n].grad.double().norm().item() - gn) <= 1e-3 * gn, n


def test_oracle_attention_with_a_per_head_mask_equals_torch_multihead_attention():
    """the (B * heads, L, L) attn_mask branch of the oracle's attention (not a form the reference uses, so no golden
    holds it): pinned against torch.nn.MultiheadAttention on the same weights, float and boolean masks, 1e-5."""
    import torch.nn as nn
    from oracle import ref_functional as RF
    torch.manual_seed(3)
    B, L, E, H = 3, 20, 32, 4
    mha = nn.MultiheadAttention(E, H, batch_first=True).eval()
    sd = {"a." + k: v for k, v in mha.state_dict().items()}
    x = torch.randn(B, L, E)
    fm = torch.randn(B * H, L, L)
    bm = torch.rand(B * H, L, L) < 0.3
    bm[:, torch.arange(L), torch.arange(L)] = False                 # no fully masked row
    for m in (fm, bm):
        with torch.no_grad():
            want, _ = mha(x, x, x, attn_mask=m, need_weights=False)
            got, _ = RF.multihead_attention(sd, "a.", x, x, H, attn_mask=m)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), (got - want).abs().max()

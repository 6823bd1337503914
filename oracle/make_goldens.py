#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by importing the
REFERENCE (read-only at /root/reference) in the build container.

The reference never travels to the GPU box; only these small fixtures do.
Weights are NOT stored: every case rebuilds them from ``torch.manual_seed`` +
``perturb_norm_state`` (below) and the fixture carries a checksum per tensor so
a mismatch in construction order / init is caught, not silently accepted.

While generating, the script also checks that
  * the product classes (multimodal_eeg_fmri_amd.*) initialise bit-identically
    to the reference classes under the same seed (same state_dict keys/values);
  * the functional oracle (oracle/ref_functional.py) reproduces the reference
    outputs to <= 2e-6.

Run:  python oracle/make_goldens.py
"""
from __future__ import annotations

import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import ref_functional as RF  # noqa: E402
from oracle.fixtures import (build, checksum, perturb_norm_state, seeded_randn,  # noqa: E402
                             grad_summary)


def _import_reference():
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):      # banners at import
        import EEG_CODE.crossmodal_v4_enhancements as cv4
        import bridge_utils as br
        import fMRI_CODE.fmri_utils as fm
    return cv4, br, fm


def _same_state(ref_mod, our_mod, what):
    a, b = ref_mod.state_dict(), our_mod.state_dict()
    assert list(a.keys()) == list(b.keys()), f"{what}: state_dict keys differ"
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), f"{what}: {k} differs"


def _close(a, b, what, tol=2e-6):
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"oracle vs reference: {what}: max|d|={err:.3e}"
    return err


def _hook_stages(model, names):
    store = {}
    handles = []
    mods = dict(model.named_modules())
    for label, modname in names.items():
        handles.append(mods[modname].register_forward_hook(
            lambda m, i, o, label=label: store.__setitem__(label, o.detach().clone())))
    return store, handles


def _np(t):
    return t.detach().cpu().numpy()


def main():
    os.makedirs(OUT, exist_ok=True)
    cv4, br, fm = _import_reference()
    import multimodal_eeg_fmri_amd.enhanced_models_v4 as ours_e
    import multimodal_eeg_fmri_amd.crossmodal_v4_enhancements as ours_c
    import multimodal_eeg_fmri_amd.fmri_utils as ours_f
    import multimodal_eeg_fmri_amd.bridge_utils as ours_b
    torch.set_num_threads(8)
    report = []

    # ---------------------------------------------------------------- (i) a3
    for tag, (B, C, T), stride in (("c1", (8, 8, 256), 4), ("c2", (2, 64, 1024), 8)):
        ref = build(cv4.EnhancedERPEncoder, 11, C, 128, 2, 4, 0.3).eval()
        our = build(ours_e.EnhancedERPEncoder, 11, C, 128, 2, 4, 0.3).eval()
        _same_state(ref, our, f"EnhancedERPEncoder[{tag}]")
        x = seeded_randn(101, B, C, T)
        store, hs = _hook_stages(ref, {"conv1": "conv_layers.2", "conv2": "conv_layers.7",
                                       "conv3": "conv_layers.11", "pos": "pos_encoder",
                                       "block0": "transformer_layers.0",
                                       "block1": "transformer_layers.1"})
        with torch.no_grad():
            y = ref(x)
        for h in hs:
            h.remove()
        st = {}
        with torch.no_grad():
            yo = RF.erp_encoder(our.state_dict(), x, stages=st)
        e = _close(yo, y, f"a3[{tag}] out")
        for k in store:
            _close(st[k], store[k], f"a3[{tag}] {k}")
        report.append(f"a3[{tag}] oracle-vs-ref max|d|={e:.2e}")
        fx = {"seed": 11, "x_seed": 101, "shape": np.array([B, C, T]), "stride": stride,
              "out": _np(y), "cks": checksum(ref)}
        for k, v in store.items():
            # conv stages are (B,C,L): subsample L; token stages are (B,L,d): subsample L
            fx["stage_" + k] = _np(v[:, :, ::stride] if k.startswith("conv") else v[:, ::stride, :])
        np.savez_compressed(os.path.join(OUT, f"a3_erp_{tag}.npz"), **fx)

    # --------------------------------------------------------------- (ii) a4
    ref = build(cv4.EnhancedPowerEncoder, 12, 64, 128, 2, 4, 0.3).eval()
    our = build(ours_e.EnhancedPowerEncoder, 12, 64, 128, 2, 4, 0.3).eval()
    _same_state(ref, our, "EnhancedPowerEncoder")
    x = seeded_randn(102, 2, 64, 256)
    with torch.no_grad():
        y = ref(x)
        yo = RF.power_encoder(our.state_dict(), x)
    report.append(f"a4 oracle-vs-ref max|d|={_close(yo, y, 'a4'):.2e}")
    np.savez_compressed(os.path.join(OUT, "a4_power.npz"), seed=12, x_seed=102,
                        shape=np.array([2, 64, 256]), out=_np(y), cks=checksum(ref))

    # -------------------------------------------------------------- (iii) a6/a7
    ref = build(cv4.EnhancedTriModalFusionNetV4Lite, 13, 8, 8, 459).eval()
    our = build(ours_c.EnhancedTriModalFusionNetV4Lite, 13, 8, 8, 459).eval()
    _same_state(ref, our, "V4Lite")
    erp, pw, conn = seeded_randn(103, 8, 8, 256), seeded_randn(104, 8, 8, 256), seeded_randn(105, 8, 459)
    with torch.no_grad():
        logits, fused = ref(erp, pw, conn, return_fused_feats=True)
        e_feat, p_feat, c_feat = ref.erp_encoder(erp), ref.pw_encoder(pw), ref.conn_encoder(conn)
        lo, fo, _ = RF.trimodal_lite(our.state_dict(), erp, pw, conn)
    _close(fo, fused, "a7 fused")
    report.append(f"a7 oracle-vs-ref max|d|={_close(lo, logits, 'a7 logits'):.2e}")
    np.savez_compressed(os.path.join(OUT, "a7_lite.npz"), seed=13, x_seeds=np.array([103, 104, 105]),
                        logits=_np(logits), fused=_np(fused), erp_feat=_np(e_feat),
                        pw_feat=_np(p_feat), conn_feat=_np(c_feat), cks=checksum(ref))

    # --------------------------------------------------------------- (iv) a9
    ref = build(fm.fMRIFusionNet, 14, 100, 200).eval()
    our = build(ours_f.fMRIFusionNet, 14, 100, 200).eval()
    _same_state(ref, our, "fMRIFusionNet")
    act, con = seeded_randn(106, 8, 100), seeded_randn(107, 8, 200)
    with torch.no_grad():
        out, fused = ref(act, con, return_features=True)
        oo, fo = RF.fmri_fusion_net(our.state_dict(), act, con)
    _close(oo, out, "a9 out")
    report.append(f"a9 oracle-vs-ref max|d|={_close(fo, fused, 'a9 fused'):.2e}")
    np.savez_compressed(os.path.join(OUT, "a9_fmri.npz"), seed=14, x_seeds=np.array([106, 107]),
                        out=_np(out), fused=_np(fused), cks=checksum(ref))

    # ---------------------------------------------------------------- (v) a11
    ref = build(br.EEGfMRIBridgeFusionNet, 15).eval()
    our = build(ours_b.EEGfMRIBridgeFusionNet, 15).eval()
    _same_state(ref, our, "EEGfMRIBridgeFusionNet")
    eeg, fmri = seeded_randn(108, 8, 128), seeded_randn(109, 8, 64)
    with torch.no_grad():
        logits, fused, fw, aw = ref(eeg, fmri, return_features=True, return_weights=True)
        ep, fp = ref.eeg_proj(eeg), ref.fmri_proj(fmri)
        lo, fo, fwo, awo = RF.bridge_net(our.state_dict(), eeg, fmri)
    for a, b, n in ((lo, logits, "logits"), (fo, fused, "fused"), (fwo, fw, "fw"), (awo, aw, "aw")):
        e = _close(a, b, "a11 " + n)
    gfw = ref.get_fusion_weights()
    report.append(f"a11 oracle-vs-ref max|d|={e:.2e}")
    np.savez_compressed(os.path.join(OUT, "a11_bridge.npz"), seed=15, x_seeds=np.array([108, 109]),
                        logits=_np(logits), fused=_np(fused), fusion_w=_np(fw), attn_w=_np(aw),
                        eeg_proj=_np(ep), fmri_proj=_np(fp),
                        gfw=np.array([gfw["eeg_weight"], gfw["fmri_weight"], gfw["temperature"]]),
                        cks=checksum(ref))

    # ---------------------------------------------------------------- (vi) a5
    fx = {}
    for M in (2, 3):
        ref = build(cv4.LearnedFusionModule, 16 + M, M, 128, perturb=False).eval()
        our = build(ours_e.LearnedFusionModule, 16 + M, M, 128, perturb=False).eval()
        with torch.no_grad():
            for mod in (ref, our):
                mod.fusion_logits.copy_(torch.linspace(0.5, 1.5, M))
                mod.temperature.fill_(0.7)
        _same_state(ref, our, f"LearnedFusion{M}")
        feats = [seeded_randn(110 + i, 8, 128) for i in range(M)]
        with torch.no_grad():
            f, w = ref(feats, return_weights=True)
            fo, wo = RF.learned_fusion(our.state_dict(), feats)
        _close(fo, f, f"a5 M={M}")
        fx[f"fused{M}"], fx[f"w{M}"] = _np(f), _np(w)
        fx[f"cks{M}"] = checksum(ref)
    np.savez_compressed(os.path.join(OUT, "a5_fusion.npz"), **fx)
    report.append("a5 ok")

    # -------------------------------------------------------------- (vii) grads
    #  train mode (BN batch statistics), constructor dropout=0.0
    ref = build(cv4.EnhancedERPEncoder, 21, 8, 128, 2, 4, 0.0).train()
    our = build(ours_e.EnhancedERPEncoder, 21, 8, 128, 2, 4, 0.0).train()
    _same_state(ref, our, "a3 train")
    x = seeded_randn(121, 8, 8, 256).requires_grad_(True)
    gy = seeded_randn(122, 8, 128)
    cks_before = checksum(ref)
    y = ref(x)
    y.backward(gy)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point())
          for k, v in our.state_dict().items()}
    xo = x.detach().clone().requires_grad_(True)
    yo = RF.erp_encoder(sd, xo, train=True)
    yo.backward(gy)
    _close(yo, y, "a3 train out", 5e-6)
    _close(xo.grad, x.grad, "a3 dx", 2e-5)
    for n, p in ref.named_parameters():
        _close(sd[n].grad, p.grad, "a3 d" + n, 5e-5)
    # cks_after pins the BatchNorm running-stat update (momentum 0.1, unbiased var)
    fx = {"seed": 21, "x_seed": 121, "gy_seed": 122, "out": _np(y), "dx": _np(x.grad),
          "cks": cks_before, "cks_after": checksum(ref)}
    fx.update(grad_summary(ref))
    np.savez_compressed(os.path.join(OUT, "a3_erp_train_grads.npz"), **fx)

    ref = build(br.EEGfMRIBridgeFusionNet, 22, dropout=0.0).train()
    our = build(ours_b.EEGfMRIBridgeFusionNet, 22, dropout=0.0).train()
    _same_state(ref, our, "a11 train")
    ref.fusion.gate_net[2].p = 0.0      # LearnedFusionModule hard-codes Dropout(0.2) (enhanced_models_v4.py:449)
    eeg = seeded_randn(123, 8, 128).requires_grad_(True)
    fmri = seeded_randn(124, 8, 64).requires_grad_(True)
    tgt = torch.tensor([0, 1, 1, 0, 1, 0, 0, 1])
    w = torch.tensor([0.7, 1.3])
    loss = torch.nn.functional.cross_entropy(ref(eeg, fmri), tgt, weight=w)
    loss.backward()
    fx = {"seed": 22, "x_seeds": np.array([123, 124]), "target": tgt.numpy(), "class_w": w.numpy(),
          "loss": np.array(loss.item()), "d_eeg": _np(eeg.grad), "d_fmri": _np(fmri.grad),
          "cks": checksum(ref)}
    fx.update(grad_summary(ref))
    np.savez_compressed(os.path.join(OUT, "a11_bridge_train_grads.npz"), **fx)
    report.append("grads ok")

    # ----------------------------------------- (f).1 full V4 classifiers (eval + train grads)
    import multimodal_eeg_fmri_amd.crossmodal_v4_enhancements as ours_c
    for tag, name, args in (("trimodal_v4", "EnhancedTriModalFusionNetV4", (8, 8, 36)),
                            ("smart_v4", "EnhancedSmartFusionNetV4", (8, 8))):
        tri = "Tri" in name
        ref = build(getattr(cv4, name), 31, *args).eval()
        our = build(getattr(ours_c, name), 31, *args).eval()
        _same_state(ref, our, "f1 " + name)
        xs = [seeded_randn(131, 4, 8, 256), seeded_randn(132, 4, 8, 256)] + ([seeded_randn(133, 4, 36)] if tri else [])
        with torch.no_grad():
            want = ref(*xs, return_fusion_weights=True, return_fused_feats=True)
            got = (RF.trimodal_v4 if tri else RF.smart_fusion_v4)(ref.state_dict(), *xs)
        for nm, w_, g_ in zip(("logits", "weights", "fused"), want, got):
            _close(g_, w_, f"f1 {name} {nm}", 5e-6)
        fx = {"seed": 31, "x_seeds": np.array([131, 132, 133]), "cks": checksum(ref),
              "logits": _np(want[0]), "weights": _np(want[1]), "fused": _np(want[2])}
        # train mode: constructor dropout 0, the fusion gate's hard-coded Dropout(0.2) off
        ref = build(getattr(cv4, name), 32, *args, dropout=0.0).train()
        ref.fusion.gate_net[2].p = 0.0
        gy = seeded_randn(134, 4, 2)
        ref(*xs).backward(gy)
        sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in ref.state_dict().items()}
        (RF.trimodal_v4 if tri else RF.smart_fusion_v4)(sd, *xs, train=True)[0].backward(gy)
        for n, pr in ref.named_parameters():
            _close(sd[n].grad, pr.grad, f"f1 {name} d{n}", 2e-4)
        fx.update({"train_seed": 32, "gy_seed": 134})
        fx.update({"t_" + k: v for k, v in grad_summary(ref, head=8).items()})
        np.savez_compressed(os.path.join(OUT, f"f1_{tag}.npz"), **fx)
    report.append("f1 V4 classifiers ok")

    # --------------------------------------------- (f).3 BalancedTriModalDataset aggregation
    def _feature_dicts():
        erp = {(s_, b_): (seeded_randn(200 + 10 * s_ + b_, 4, 6), {"band": b_}) for s_ in (1, 2, 3, 5) for b_ in range(3)}
        pw = {(s_, b_): seeded_randn(300 + 10 * s_ + b_, 4, 6).numpy() for s_ in (1, 2, 3, 4) for b_ in range(2)}
        conn = {s_: seeded_randn(400 + s_, 5, 5) for s_ in (1, 2, 3, 5, 6)}
        labels = {1: 0, 2: 1, 3: 1, 4: 0, 6: 1}
        return erp, pw, conn, labels
    fx = {}
    for method in ("mean", "max", "first"):
        with contextlib.redirect_stdout(io.StringIO()):
            ref_ds = cv4.BalancedTriModalDataset(*_feature_dicts(), agg_method=method)
            our_ds = ours_c.BalancedTriModalDataset(*_feature_dicts(), agg_method=method)
        assert len(ref_ds) == len(our_ds) == 3
        for i in range(len(ref_ds)):
            a, b = ref_ds[i], our_ds[i]
            assert a[3:] == b[3:] and all(torch.equal(x, y) for x, y in zip(a[:3], b[:3])), (method, i)
            fx[f"{method}_{i}_erp"], fx[f"{method}_{i}_pw"], fx[f"{method}_{i}_conn"] = _np(a[0]), _np(a[1]), _np(a[2])
            fx[f"{method}_{i}_meta"] = np.array([a[3], a[4]])
    np.savez_compressed(os.path.join(OUT, "f3_balanced_dataset.npz"), **fx)
    report.append("f3 BalancedTriModalDataset ok")

    # --------------------------------------------- (f).3 on-disk loaders (CSV / MATLAB-v5 trees)
    import json as _json
    import tempfile
    import types
    import pandas as pd
    from scipy.io import savemat
    import multimodal_eeg_fmri_amd.eeg_data_utils as ours_ed
    rng = np.random.default_rng(7)

    def _nan(a, frac=0.1):
        a = a.astype(np.float32)
        a[rng.random(a.shape) < frac] = np.nan
        return a
    csv_files, mat_files = {}, {}
    for subj in (1, 2, 3):
        for t in ("faces", "tools"):
            if subj == 3 and t == "tools":
                continue                                   # a missing activation type
            df = pd.DataFrame(_nan(rng.standard_normal((5, 6))), columns=[f"roi{i}" for i in range(6)])
            if subj != 2:
                df.insert(0, "Subject", subj)
            csv_files[f"fmri/sub-{subj}/subject_{subj}_activation_{t}.csv"] = df.to_csv(index=False)
        df = pd.DataFrame(_nan(rng.standard_normal((4, 4))), columns=[f"r{i}" for i in range(4)])
        csv_files[f"fmri/sub-{subj}/subject_{subj}_fdr_PPI_Connectivity_rest.csv"] = df.to_csv(index=False)
    csv_files["fmri/labels/outcomes.csv"] = "ID,Outcome,extra\n1,good,a\n2,Bad,b\n3,YES,c\n9,good,d\n"
    csv_files["eeg/labels/medical_score.csv"] = ("Subject,Postoperative evaluation\nsub01,1\nsub02,3\nsub03,\n"
                                                 "sub04,2\nsub05,4\n")
    mat_files["eeg/conn/conn_Alpha_open_sub01.mat"] = _nan(rng.standard_normal((5, 5)))
    mat_files["eeg/conn/conn_beta_close_sub02.mat"] = _nan(rng.standard_normal((5, 5)))     # band-key fallback name
    mat_files["eeg/pw/powspctrm_alpha_1_Hz_sub01.mat"] = _nan(rng.standard_normal((8, 3)))
    mat_files["eeg/pw/powspctrm_beta_2_Hz_sub02.mat"] = _nan(rng.standard_normal((8, 3)))
    mat_files["eeg/erp/ERP_sub01_alpha_1_Hz_avg.mat"] = _nan(rng.standard_normal((8, 20)))
    mat_files["eeg/erp/ERP_sub02_beta_2_Hz.mat"] = _nan(rng.standard_normal((8, 20)))
    if "h5py" not in sys.modules:
        # h5py is not installed here; the reference module imports it at the top.  An EMPTY module object
        # lets the import succeed; every h5py.* use then raises and the reference takes its own
        # scipy.loadmat fallback branch (eeg_data_utils.py:168-180).  Its HDF5 branch stays unpinned.
        sys.modules["h5py"] = types.ModuleType("h5py")
    with contextlib.redirect_stdout(io.StringIO()):
        import EEG_CODE.eeg_data_utils as ed
    with tempfile.TemporaryDirectory() as tmp:
        for rel, text in csv_files.items():
            os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
            with open(os.path.join(tmp, rel), "w") as fh:
                fh.write(text)
        for rel, arr in mat_files.items():
            os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
            savemat(os.path.join(tmp, rel), {"data": arr})
        subs = [1, 2, 3, 5]
        expected = {}

        def _cmp(name, ref_d, our_d):
            assert list(ref_d.keys()) == list(our_d.keys()), (name, list(ref_d), list(our_d))
            for k in ref_d:
                a, b = np.asarray(ref_d[k]), np.asarray(our_d[k])
                assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b), (name, k)
                expected[f"{name}|{_json.dumps(k)}"] = a
        f = os.path.join(tmp, "fmri")
        for agg in ("mean", "std", "both"):
            _cmp(f"act_{agg}", fm.load_activation_features(f, subs, ["faces", "tools"], agg),
                 ours_f.load_activation_features(f, subs, ["faces", "tools"], agg))
        assert fm.load_activation_features(f, subs, ["faces"], "median") == \
            ours_f.load_activation_features(f, subs, ["faces"], "median") == {}
        _cmp("conn", fm.load_connectivity_features(f, subs, ["rest", "task"]),
             ours_f.load_connectivity_features(f, subs, ["rest", "task"]))
        lr_, lo_ = fm.load_fmri_labels(os.path.join(f, "labels"), subs), ours_f.load_fmri_labels(os.path.join(f, "labels"), subs)
        assert lr_ == lo_ == {1: 1, 2: 0, 3: 1}
        e = os.path.join(tmp, "eeg")
        for binary in (True, False):
            a, b = ed.load_eeg_labels(os.path.join(e, "labels"), binary), ours_ed.load_eeg_labels(os.path.join(e, "labels"), binary)
            assert a == b, (a, b)
            expected[f"eeg_labels_{int(binary)}"] = np.array(sorted(a.items()), dtype=np.float64)
        bands = {"alpha": "Alpha", "beta": "Beta"}
        _cmp("eeg_conn", ed.load_eeg_conn_features(os.path.join(e, "conn"), subs, bands, ["open", "close"]),
             ours_ed.load_eeg_conn_features(os.path.join(e, "conn"), subs, bands, ["open", "close"]))
        _cmp("eeg_pw", ed.load_eeg_pw_features(os.path.join(e, "pw"), subs, ["alpha", "beta"], ["1_Hz", "2_Hz"]),
             ours_ed.load_eeg_pw_features(os.path.join(e, "pw"), subs, ["alpha", "beta"], ["1_Hz", "2_Hz"]))
        _cmp("eeg_erp", ed.load_eeg_erp_features(os.path.join(e, "erp"), subs, ["alpha", "beta"], ["1_Hz", "2_Hz"]),
             ours_ed.load_eeg_erp_features(os.path.join(e, "erp"), subs, ["alpha", "beta"], ["1_Hz", "2_Hz"]))
    fx = {"csv_paths": np.array(list(csv_files)), "csv_texts": np.array(list(csv_files.values())),
          "mat_paths": np.array(list(mat_files)), "expected_keys": np.array(list(expected))}
    for i, arr in enumerate(mat_files.values()):
        fx[f"mat_{i}"] = arr
    for i, arr in enumerate(expected.values()):
        fx[f"exp_{i}"] = arr
    np.savez_compressed(os.path.join(OUT, "f3_loaders.npz"), **fx)
    report.append(f"f3 loaders ok ({len(expected)} entries; reference HDF5 branch unpinned: h5py absent)")

    # ------------------- (f).1/(f).3 notebook classes: PerFoldNormalizer, FocalLoss, collate_trimodal
    nb = _json.load(open(os.path.join(REF, "EEG_CODE", "CrossModal_EEG_scr.ipynb")))
    ns = {"np": np, "torch": torch, "nn": torch.nn, "F": torch.nn.functional}
    for cell in (19, 20, 24):                        # the notebook is not a module: run the three cells
        with contextlib.redirect_stdout(io.StringIO()):
            exec("".join(nb["cells"][cell]["source"]), ns)
    import multimodal_eeg_fmri_amd.crossmodal_eeg_scr as ours_nb
    data = {(s_, b_, 0): rng.standard_normal((3, 4)).astype(np.float32) * (1 + s_) for s_ in (1, 2, 3, 4) for b_ in range(2)}
    subj_arr, train_idx = np.array([1, 2, 3, 4]), np.array([0, 2, 3])
    with contextlib.redirect_stdout(io.StringIO()):
        rn, on = ns["PerFoldNormalizer"](), ours_nb.PerFoldNormalizer()
        rn.fit_on_indices(data, train_idx, subj_arr); on.fit_on_indices(data, train_idx, subj_arr)
    assert rn.stats["mean"] == on.stats["mean"] and rn.stats["std"] == on.stats["std"]
    rt, ot = rn.transform(data), on.transform(data)
    assert all(np.array_equal(rt[k], ot[k]) for k in rt)
    fx = {"norm_keys": np.array(list(data)), "norm_vals": np.stack(list(data.values())), "norm_train_idx": train_idx,
          "norm_subjects": subj_arr, "norm_mean": np.array(rn.stats["mean"]), "norm_std": np.array(rn.stats["std"]),
          "norm_out": np.stack(list(rt.values()))}
    logits, tgt = seeded_randn(127, 16, 3) * 2.0, torch.randint(0, 3, (16,), generator=torch.Generator().manual_seed(128))
    fx["focal_logits"], fx["focal_target"] = _np(logits), tgt.numpy()
    for alpha, gamma in ((0.25, 2.0), (1.0, 0.0), (0.5, 1.5)):
        for red in ("mean", "sum", "none"):
            z = logits.clone().requires_grad_(True)
            loss = ns["FocalLoss"](alpha, gamma, red)(z, tgt)
            (loss.sum() * 1.0).backward()
            fx[f"focal_{alpha}_{gamma}_{red}"], fx[f"focal_{alpha}_{gamma}_{red}_grad"] = _np(loss), _np(z.grad)
    batch5 = [(seeded_randn(130 + i, 20, 4), seeded_randn(140 + i, 3, 12), seeded_randn(150 + i, 7), i, i % 2) for i in range(3)]
    batch4 = [b[:2] + b[3:] for b in batch5]
    for b_ in (batch5, batch4):
        r_, o_ = ns["collate_trimodal"](b_), ours_nb.collate_trimodal(b_)
        assert all((x is None and y is None) or torch.equal(x, y) for x, y in zip(r_, o_))
    fx["collate_erp_shape"] = np.array(ns["collate_trimodal"](batch5)[0].shape)
    np.savez_compressed(os.path.join(OUT, "f3_notebook_classes.npz"), **fx)
    report.append("f3 PerFoldNormalizer / FocalLoss / collate_trimodal ok")

    # ------------------------------------------------------------- (viii) a8
    logits, tgt = seeded_randn(125, 16, 2), (seeded_randn(126, 16) > 0).long()
    ls = cv4.LabelSmoothingCrossEntropy(0.1)(logits, tgt)
    _close(RF.label_smoothing_ce(logits, tgt, 0.1), ls, "a8 loss")
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=5e-5)
    sch = cv4.CosineAnnealingWarmup(opt, warmup_epochs=3, total_epochs=50)
    lrs = np.array([sch.step() for _ in range(50)])
    es = cv4.EarlyStopping(patience=3, mode="max")
    scores = [0.5, 0.52, 0.5205, 0.51, 0.53, 0.5, 0.5, 0.5, 0.5]
    stops = np.array([bool(es(s)) for s in scores])
    np.savez_compressed(os.path.join(OUT, "a8_train_utils.npz"), logits=_np(logits), target=tgt.numpy(),
                        ls_loss=np.array(ls.item()), lrs=lrs, es_scores=np.array(scores), es_stops=stops)
    report.append("a8 ok")

    # ------------------------------------------------ state_dict key/shape layout
    import json
    layout = {}
    for name, mod in (("EnhancedERPEncoder", cv4.EnhancedERPEncoder(64)),
                      ("EnhancedPowerEncoder", cv4.EnhancedPowerEncoder(64)),
                      ("LearnedFusionModule", cv4.LearnedFusionModule(3, 128)),
                      ("EnhancedTriModalFusionNetV4Lite", cv4.EnhancedTriModalFusionNetV4Lite(8, 8, 459)),
                      ("EnhancedTriModalFusionNetV4", cv4.EnhancedTriModalFusionNetV4(64, 64, 459)),
                      ("EnhancedSmartFusionNetV4", cv4.EnhancedSmartFusionNetV4(64, 64)),
                      ("fMRIFusionNet", fm.fMRIFusionNet(100, 200)),
                      ("EEGfMRIBridgeFusionNet", br.EEGfMRIBridgeFusionNet())):
        layout[name] = [[k, list(v.shape)] for k, v in mod.state_dict().items()]
    with open(os.path.join(OUT, "state_dict_layout.json"), "w") as fh:
        json.dump(layout, fh, indent=0)

    print("\n".join(report))
    print("fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()

"""CPU: the C-ABI library loads and exports every symbol include/mmeeg_hip.h
declares (no compute without a GPU), the product path refuses CPU tensors, and
the host-side logic (config, datasets, collate, class surface) behaves like the
reference's."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from multimodal_eeg_fmri_amd import _hip
import multimodal_eeg_fmri_amd.bridge_utils as B
import multimodal_eeg_fmri_amd.crossmodal_v4_enhancements as C
import multimodal_eeg_fmri_amd.enhanced_models_v4 as E
import multimodal_eeg_fmri_amd.fmri_utils as Fm
from multimodal_eeg_fmri_amd.config import Config, set_seed

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_library_exports_every_declared_symbol():
    sigs = _hip.parse_header()
    assert len(sigs) >= 40
    assert os.path.exists(_hip.lib_path()), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_hip.lib_path())
    missing = [n for n in list(sigs) + ["mm_last_error", "mm_abi_version"] if not hasattr(lib, n)]
    assert not missing, missing
    lib.mm_abi_version.restype = ctypes.c_int
    assert lib.mm_abi_version() == 1


def test_argument_errors_are_reported_not_crashed():
    """entry points validate before touching the device: callable without a GPU."""
    lib = _hip.load()
    rc = lib.mm_pack_nct_bf16(None, None, 0, 0, 0, 0, None)
    assert rc == -1 and b"pack_nct" in lib.mm_last_error()
    rc = lib.mm_attn_fwd(ctypes.c_void_p(8), ctypes.c_void_p(8), None, 1, 16, 4, 64, ctypes.c_float(0.1),
                         ctypes.c_float(0.0), 0, None, None)
    assert rc == -1 and b"head_dim" in lib.mm_last_error()


def test_product_path_has_no_cpu_fallback():
    m = E.EnhancedERPEncoder(8).eval()
    with pytest.raises(_hip.HipLibraryError, match="CPU tensor"):
        m(torch.randn(2, 8, 64))
    with pytest.raises(_hip.HipLibraryError):
        B.EEGfMRIBridgeFusionNet().eval()(torch.randn(2, 128), torch.randn(2, 64))
    src = open(os.path.join(os.path.dirname(_hip.__file__), "ops.py")).read()
    assert "oracle" not in src.replace("CPU oracle", "")


def test_state_dict_layout_matches_reference():
    """key names + shapes captured from the reference classes (oracle/make_goldens.py)"""
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_layout.json")))
    mk = {"EnhancedERPEncoder": lambda: E.EnhancedERPEncoder(64),
          "EnhancedPowerEncoder": lambda: E.EnhancedPowerEncoder(64),
          "LearnedFusionModule": lambda: E.LearnedFusionModule(3, 128),
          "EnhancedTriModalFusionNetV4Lite": lambda: C.EnhancedTriModalFusionNetV4Lite(8, 8, 459),
          "EnhancedTriModalFusionNetV4": lambda: C.EnhancedTriModalFusionNetV4(64, 64, 459),
          "EnhancedSmartFusionNetV4": lambda: C.EnhancedSmartFusionNetV4(64, 64),
          "fMRIFusionNet": lambda: Fm.fMRIFusionNet(100, 200),
          "EEGfMRIBridgeFusionNet": lambda: B.EEGfMRIBridgeFusionNet()}
    for name, layout in ref.items():
        sd = mk[name]().state_dict()
        assert [[k, list(v.shape)] for k, v in sd.items()] == layout, name


def test_config_surface_and_yaml_roundtrip(tmp_path):
    cfg = Config(None, make_dirs=False)
    assert cfg.batch_size == 8 and cfg.num_epochs == 50 and cfg.learning_rate == 5e-5
    assert cfg.grad_clip == 1.0 and cfg.n_splits == 5 and len(cfg.subject_list) == 63
    assert cfg.freq_bands is cfg.eeg_segments and cfg.epochs == cfg.num_epochs   # aliases main() reads
    cfg.batch_size = 16
    p = tmp_path / "c.yaml"
    cfg.save_config(str(p))
    cfg2 = Config(str(p), make_dirs=False)
    assert cfg2.batch_size == 16
    (tmp_path / "d.yaml").write_text("batch_size: 4\nnot_an_attr: 1\n")
    cfg3 = Config(str(tmp_path / "d.yaml"), make_dirs=False)
    assert cfg3.batch_size == 4 and not hasattr(cfg3, "not_an_attr")
    set_seed(7)
    a = torch.rand(3)
    set_seed(7)
    assert torch.equal(a, torch.rand(3))


def test_bridge_dataset_alignment_and_collate():
    eeg = {"001": torch.ones(128), 2: torch.full((128,), 2.0), 5: torch.zeros(128)}
    fmri = {1: torch.ones(64), "2": torch.full((64,), 2.0), 9: torch.zeros(64)}
    labels = {1: 0, 2: 1, 5: 1, 9: 0}
    ds = B.BridgeFeatureDataset(eeg, fmri, labels, ["2", 1, 5, 9])
    assert len(ds) == 2 and [s[3] for s in ds] == [1, 2]
    e, f, y, subj = B.collate_bridge([ds[0], ds[1]])
    assert e.shape == (2, 128) and f.shape == (2, 64) and y.dtype == torch.long and subj == [1, 2]
    assert len(B.BridgeFeatureDataset({}, {}, {}, [])) == 0


def test_lite_wrapper_and_collate_surface():
    import multimodal_eeg_fmri_amd.run_training_lite as R
    for name in ("main", "ImprovedTriModalFusionNetLite", "collate_balanced", "aggregate_features",
                 "load_labels", "normalize_modality", "vec_upper_triangle"):
        assert hasattr(R, name), name
    batch = [{"erp": torch.zeros(8, 16), "pw": torch.zeros(8, 16), "conn": torch.zeros(10), "label": 1, "subject": 3},
             (torch.ones(8, 16), torch.ones(8, 16), torch.ones(10), 0, 4)]
    erp, pw, conn, y, subj = R.collate_balanced(batch)
    assert erp.shape == (2, 8, 16) and conn.shape == (2, 10) and y.tolist() == [1, 0] and subj == [3, 4]
    m = np.arange(16.0).reshape(4, 4)
    assert R.vec_upper_triangle(m).tolist() == [1, 2, 3, 6, 7, 11]
    z = R.normalize_modality(np.array([1.0, 2.0, 3.0]))
    assert abs(z.mean()) < 1e-9


def test_f3_balanced_trimodal_dataset_matches_reference_golden():
    """per-subject aggregation of BalancedTriModalDataset (tuple keys, (feature, metadata)
    values, numpy / tensor entries, mean / max / first) against the reference's own output."""
    import contextlib, io
    import numpy as np
    from oracle.fixtures import seeded_randn
    fx = np.load(os.path.join(GOLDEN, "f3_balanced_dataset.npz"))
    erp = {(s, b): (seeded_randn(200 + 10 * s + b, 4, 6), {"band": b}) for s in (1, 2, 3, 5) for b in range(3)}
    pw = {(s, b): seeded_randn(300 + 10 * s + b, 4, 6).numpy() for s in (1, 2, 3, 4) for b in range(2)}
    conn = {s: seeded_randn(400 + s, 5, 5) for s in (1, 2, 3, 5, 6)}
    labels = {1: 0, 2: 1, 3: 1, 4: 0, 6: 1}
    for method in ("mean", "max", "first"):
        with contextlib.redirect_stdout(io.StringIO()):
            ds = C.BalancedTriModalDataset(erp, pw, conn, labels, agg_method=method)
        assert len(ds) == 3
        for i in range(3):
            e, p, c, lab, subj = ds[i]
            np.testing.assert_array_equal(e.numpy(), fx[f"{method}_{i}_erp"])
            np.testing.assert_array_equal(p.numpy(), fx[f"{method}_{i}_pw"])
            np.testing.assert_array_equal(c.numpy(), fx[f"{method}_{i}_conn"])
            assert [lab, subj] == list(fx[f"{method}_{i}_meta"])
    with contextlib.redirect_stdout(io.StringIO()):
        ds = C.BalancedTriModalDataset(erp, pw, conn, labels, transform=lambda t: t * 2)
    assert torch.equal(ds[0][0], torch.as_tensor(fx["mean_0_erp"]) * 2) and torch.equal(ds[0][2], torch.as_tensor(fx["mean_0_conn"]))

"""CPU: the C-ABI library loads and exports every symbol include/mmeeg_hip.h
declares (no compute without a GPU), the product path refuses CPU tensors, and
the host-side logic (config, datasets, collate, class surface) behaves like the
reference's."""
import ctypes
import json
import os
import sys

import numpy as np
import pytest
import torch

from multimodal_eeg_fmri_amd import _hip
import multimodal_eeg_fmri_amd.bridge_utils as B
import multimodal_eeg_fmri_amd.crossmodal_v4_enhancements as C
import multimodal_eeg_fmri_amd.enhanced_models_v4 as E
import multimodal_eeg_fmri_amd.fmri_utils as Fm
from multimodal_eeg_fmri_amd.config import Config, set_seed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_library_exports_every_declared_symbol():
    sigs = _hip.parse_header()
    assert len(sigs) >= 40
    assert os.path.exists(_hip.lib_path()), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_hip.lib_path())
    missing = [n for n in list(sigs) + ["mm_last_error", "mm_abi_version"] if not hasattr(lib, n)]
    assert not missing, missing
    lib.mm_abi_version.restype = ctypes.c_int
    assert lib.mm_abi_version() == 5


def test_argument_errors_are_reported_not_crashed():
    """entry points validate before touching the device: callable without a GPU."""
    lib = _hip.load()
    rc = lib.mm_pack_nct_bf16(None, None, 0, 0, 0, 0, None)
    assert rc == -1 and b"pack_nct" in lib.mm_last_error()
    rc = lib.mm_attn_fwd(ctypes.c_void_p(8), ctypes.c_void_p(8), None, 1, 16, 4, 64, ctypes.c_float(0.1),
                         ctypes.c_float(0.0), 0, None, None, 0, None)
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


def _rebuild_tree(fx, root):
    from scipy.io import savemat
    for rel, text in zip(fx["csv_paths"], fx["csv_texts"]):
        path = os.path.join(root, str(rel))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            fh.write(str(text))
    for i, rel in enumerate(fx["mat_paths"]):
        path = os.path.join(root, str(rel))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        savemat(path, {"data": fx[f"mat_{i}"]})


def test_f3_csv_and_mat_loaders_match_reference_golden(golden, tmp_path):
    """the fMRI CSV loaders and the EEG .mat/CSV loaders return exactly (bit-equal, same
    keys in the same order) what the reference's loaders returned for the same synthetic tree
    (oracle/make_goldens.py wrote the tree and ran both).  The tree holds NaNs, a 'Subject' column in
    some files, a missing activation type, a subject without a directory, the band-key fallback name.
    UNPINNED: the HDF5 (MATLAB v7.3) branch of load_eeg_erp_features (reference eeg_data_utils.py:140-165) - h5py is not in
    the image, so neither the reference (imported with an empty `h5py` module object) nor this test can take it; the
    MATLAB-v5 files here go through both sides' scipy.io.loadmat fallback."""
    import multimodal_eeg_fmri_amd.eeg_data_utils as Ed
    fx = golden("f3_loaders.npz")
    _rebuild_tree(fx, str(tmp_path))
    subs = [1, 2, 3, 5]
    f, e = str(tmp_path / "fmri"), str(tmp_path / "eeg")
    got = {}

    def put(name, d):
        for k, v in d.items():
            got[f"{name}|{json.dumps(k)}"] = np.asarray(v)
    for agg in ("mean", "std", "both"):
        put(f"act_{agg}", Fm.load_activation_features(f, subs, ["faces", "tools"], agg))
    assert Fm.load_activation_features(f, subs, ["faces"], "median") == {}     # the reference warns and skips
    put("conn", Fm.load_connectivity_features(f, subs, ["rest", "task"]))
    assert Fm.load_fmri_labels(os.path.join(f, "labels"), subs) == {1: 1, 2: 0, 3: 1}
    for binary in (True, False):
        lab = Ed.load_eeg_labels(os.path.join(e, "labels"), binary)
        got[f"eeg_labels_{int(binary)}"] = np.array(sorted(lab.items()), dtype=np.float64)
    bands = {"alpha": "Alpha", "beta": "Beta"}
    put("eeg_conn", Ed.load_eeg_conn_features(os.path.join(e, "conn"), subs, bands, ["open", "close"]))
    put("eeg_pw", Ed.load_eeg_pw_features(os.path.join(e, "pw"), subs, ["alpha", "beta"], ["1_Hz", "2_Hz"]))
    put("eeg_erp", Ed.load_eeg_erp_features(os.path.join(e, "erp"), subs, ["alpha", "beta"], ["1_Hz", "2_Hz"]))
    keys = [str(k) for k in fx["expected_keys"]]
    assert list(got) == keys
    for i, k in enumerate(keys):
        exp = fx[f"exp_{i}"]
        assert got[k].shape == exp.shape and got[k].dtype == exp.dtype and np.array_equal(got[k], exp), k
    with pytest.raises(FileNotFoundError):
        Ed.load_eeg_labels(str(tmp_path / "nowhere"))
    with pytest.raises(ValueError):
        (tmp_path / "bad").mkdir()
        (tmp_path / "bad" / "labels.csv").write_text("who,what\n1,2\n")
        Fm.load_fmri_labels(str(tmp_path / "bad"), subs)


def test_f3_per_fold_normalizer_and_collate_match_reference_golden(golden):
    import multimodal_eeg_fmri_amd.crossmodal_eeg_scr as Nb
    fx = golden("f3_notebook_classes.npz")
    data = {tuple(int(v) for v in k): a for k, a in zip(fx["norm_keys"], fx["norm_vals"])}
    n = Nb.PerFoldNormalizer()
    n.fit_on_indices(data, fx["norm_train_idx"], fx["norm_subjects"])
    assert n.stats["mean"] == fx["norm_mean"] and n.stats["std"] == fx["norm_std"]
    out = n.transform(data)
    assert np.array_equal(np.stack(list(out.values())), fx["norm_out"])
    batch = [(torch.randn(20, 4), torch.randn(3, 12), torch.randn(7), i, i % 2) for i in range(3)]
    erp, pw, conn, subj, y = Nb.collate_trimodal(batch)
    assert list(erp.shape) == list(fx["collate_erp_shape"]) == [3, 4, 20] and pw.shape == (3, 3, 12)
    assert conn.shape == (3, 7) and subj.dtype == y.dtype == torch.long
    assert Nb.collate_trimodal([b[:2] + b[3:] for b in batch])[2] is None
    with pytest.raises(ValueError):
        Nb.collate_trimodal([b[:3] for b in batch])


def test_a10_fmri_dataset_collate_and_metrics():
    act = {s: torch.full((4,), float(s)) for s in (3, 1, 2, 7)}
    conn = {s: torch.full((6,), float(s)) for s in (1, 2, 3)}
    ds = Fm.fMRIDataset(act, conn, {1: 0, 2: 1, 3: 1, 9: 0}, reg_labels={2: 0.5}, transform=lambda t: t * 2)
    assert [s["subject"] for s in ds.samples] == [1, 2, 3] and len(ds) == 3
    a, c, y, r, subj = Fm.collate_fmri([ds[i] for i in range(3)])
    assert a.shape == (3, 4) and c.shape == (3, 6) and a[2, 0] == 6.0
    assert y.tolist() == [0, 1, 1] and r.tolist() == [0.0, 0.5, 0.0] and subj == [1, 2, 3]
    m = Fm.classification_metrics(np.array([0, 1, 1, 0]), np.array([0, 1, 0, 0]),
                                  np.array([[.9, .1], [.2, .8], [.6, .4], [.7, .3]]))
    assert m["Accuracy"] == 0.75 and m["AUC"] == 1.0 and set(m) == {"Accuracy", "F1", "Precision", "Recall", "AUC"}
    w = B.balanced_class_weights([0, 0, 0, 1])
    assert torch.allclose(w, torch.tensor([4 / 6, 4 / 2]))


def test_fused_adamw_state_dict_is_torch_adamw_layout():
    """checkpoint drop-in (FlexibleTrainer.save_checkpoint, EEG notebook cell 23): torch.optim.AdamW loads
    the FusedAdamW state_dict and the other way round."""
    from multimodal_eeg_fmri_amd.optim import FusedAdamW
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    net[0].bias.requires_grad_(False)
    fused = FusedAdamW(net.parameters(), lr=3e-4, weight_decay=0.02, betas=(0.8, 0.95))
    fused.bucket.m.copy_(torch.arange(fused.bucket.n, dtype=torch.float32))
    fused.bucket.v.fill_(2.0)
    fused.bucket.state[0] = 7.0
    sd = fused.state_dict()
    ref = torch.optim.AdamW(net.parameters())
    assert set(sd["param_groups"][0]) == set(ref.state_dict()["param_groups"][0])
    ref.load_state_dict(sd)
    g = ref.param_groups[0]
    assert g["lr"] == 3e-4 and g["weight_decay"] == 0.02 and tuple(g["betas"]) == (0.8, 0.95)
    assert 1 not in ref.state_dict()["state"]                     # the frozen bias carries no state
    st = ref.state[net[1].weight]
    assert float(st["step"]) == 7.0 and torch.equal(st["exp_avg_sq"], torch.full((2, 3), 2.0))
    assert torch.equal(ref.state[net[0].weight]["exp_avg"], torch.arange(12.0).view(3, 4))
    fused2 = FusedAdamW(net.parameters())
    fused2.load_state_dict(ref.state_dict())
    assert torch.equal(fused2.bucket.m, fused.bucket.m) and torch.equal(fused2.bucket.v, fused.bucket.v)
    assert fused2.bucket.state[0] == 7.0 and fused2.param_groups[0]["lr"] == 3e-4 and fused2.betas == (0.8, 0.95)


def test_every_mm_name_used_by_the_package_is_declared_in_the_header():
    """the ctypes signatures are parsed from include/mmeeg_hip.h, so a declared-but-missing symbol is caught at
    load time; this closes the other direction: a call to a name that was never declared (round 1:
    ops.add_positional -> "mm_add_pe") must fail here, on the CPU, not on first use on the GPU."""
    import glob
    import re
    declared = set(_hip.parse_header()) | {"mm_last_error", "mm_abi_version"}
    pkg = os.path.dirname(_hip.__file__)
    root = os.path.dirname(pkg)
    used = {}
    for path in glob.glob(os.path.join(pkg, "*.py")) + glob.glob(os.path.join(root, "tools", "*.py")) + \
            [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]:
        for name in re.findall(r"[\"'](mm_[a-z0-9_]+)[\"']", open(path).read()):
            used.setdefault(name, path)
    used.pop("mm_name", None)                      # _hip.py's own docstring example
    missing = {n: p for n, p in used.items() if n not in declared}
    assert not missing, f"called but not declared in include/mmeeg_hip.h: {missing}"
    lib = _hip.load()
    assert all(hasattr(lib, n) for n in declared)


def test_hand_scheduled_conv3d_kernel_keeps_the_accumulator_file_to_itself():
    """csrc/conv3d_wres.hip keeps accumulators, fragments, the prefetched halo and per-lane constants in literal
    AGPRs a0-a251 across inline-asm statements.  That is only sound while the COMPILER never touches those
    registers between the statements (it does not know they are live): audit the compiled ISA - no accumulator-file
    instruction or operand outside the ;;#ASMSTART / ;;#ASMEND blocks, no scratch, no spills - and that the
    committed instruction streams are what tools/gen_wres_asm.py generates."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(_hip.__file__))
    csrc = os.path.join(root, "multimodal_eeg_fmri_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        inc = os.path.join(tmp, "gen.inc")
        subprocess.run([sys.executable, os.path.join(root, "tools", "gen_wres_asm.py"), inc], check=True, capture_output=True,
                       env={k: v for k, v in os.environ.items() if not k.startswith("WRES_")})
        assert open(inc).read() == open(os.path.join(csrc, "conv3d_wres_asm.inc")).read(), "regenerate conv3d_wres_asm.inc"
        out = os.path.join(tmp, "wres.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{csrc}",
                        f"-I{os.path.join(root, 'include')}", "-S", "--cuda-device-only", "-o", out,
                        os.path.join(csrc, "conv3d_wres.hip")], check=True, capture_output=True)
        text = open(out).read()
    import re
    assert re.search(r"; ScratchSize: 0\b", text) and ".vgpr_spill_count: 0" in text.replace("    ", " ")
    inside, bad = False, []
    for line in text.splitlines():
        if "#ASMSTART" in line:
            inside = True
        elif "#ASMEND" in line:
            inside = False
        elif not inside and not line.lstrip().startswith((";", ".")) and re.search(r"\bv_accvgpr|\ba\[?\d", line):
            bad.append(line.strip())
    assert not bad, bad[:5]
    assert text.count("#ASMSTART") >= 10


def test_lite_disk_datasets_vs_reference_golden(tmp_path):
    """EEGDatasetERP / EEGDatasetPW / EEGDatasetCONN + aggregate_features (run_training_lite.py:62-259): every item tuple
    and every per-subject aggregate bit-equal to what the REFERENCE's classes produced on the same MATLAB-v5 tree
    (tests/golden/f3_lite_datasets.npz, written by oracle/make_goldens_r4.py; the tree is rebuilt from the fixture).
    Covers: variable-name search and the first-variable fallback, matrix -> strict upper triangle vs flatten, the second
    glob pattern (and the duplicates it yields), the "1_Hz" / "11_Hz" substring match, an unreadable file, an absent
    subject, labels missing for a subject.  The HDF5 (MATLAB v7.3) branch is UNPINNED: h5py is not in the image."""
    from oracle.make_goldens_r4 import ARGS, flatten, run_datasets, write_tree
    import multimodal_eeg_fmri_amd.run_training_lite as R
    fx = np.load(os.path.join(GOLDEN, "f3_lite_datasets.npz"), allow_pickle=False)
    tree = {}
    for i, rel in enumerate(fx["paths"]):
        var = str(fx[f"file_{i}_var"])
        tree[str(rel)] = {var: fx[f"file_{i}_arr"]} if var else None
    write_tree(str(tmp_path), tree)
    got = flatten(*run_datasets(R, str(tmp_path)))
    keys = [str(k) for k in fx["expected_keys"]]
    assert list(got) == keys
    for i, k in enumerate(keys):
        want = fx[f"exp_{i}"]
        assert got[k].shape == want.shape and got[k].dtype == want.dtype and np.array_equal(got[k], want), k
    # the glue of main(): aggregates -> one 2-D sample per subject present everywhere (subject 4 has no files)
    ds = R.EEGDatasetERP(ARGS["subjects"], ARGS["bands"], ARGS["freqs"], tmp_path / "erp", labels=ARGS["labels"])
    item = ds[0]
    assert isinstance(item[0], torch.Tensor) and item[0].dtype == torch.float32 and item[0].shape == (4, 16) and len(item) == 5
    agg = {k: R.aggregate_features(d, k)[0] for k, d in run_datasets(R, str(tmp_path))[0].items()}
    tri = R.AggregatedTriModalDataset(agg["erp"], agg["pw"], agg["conn"], ARGS["labels"])
    assert [s["subject"] for s in tri.samples] == [1, 2, 3] and tri[0]["erp"].shape == (4, 16) and tri[0]["conn"].shape == (15,)


def test_built_library_has_no_half_selecting_packed_fp32():
    """profiles/r03_packed_fp32_ab.txt / r04_packed_fp32_isa.txt: compiler-formed `v_pk_{fma,mul,add}_f32` whose `op_sel:`
    modifier makes the LOW result lane read the HIGH register of an operand pair (`v_pk_fma_f32 v[28:29], v[28:29],
    v[22:23], v[18:19] op_sel:[0,1,1]`: the BatchNorm scale / shift of an odd column broadcast from the high half) gave
    run-to-run different sums inside the two-stream training graph.  The guard used to be a build flag alone
    (-fno-slp-vectorize); this audits what was actually BUILT: every gfx950 code object of libmmeeg_hip.so is
    disassembled and no packed-fp32 instruction may carry an `op_sel:` modifier (`op_sel_hi:` - the broadcast of an inline
    constant or an SGPR, e.g. `v_pk_add_f32 v[0:1], v[2:3], 0 op_sel_hi:[1,0]` - is a different encoding and stays
    allowed, as do the plain register-pair `v_pk_add_f32` of conv3d_wres's hand-written K loop)."""
    import re
    import shutil
    import subprocess
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    assert os.path.exists(_hip.lib_path()), "run __graft_entry__.build() first"
    with tempfile.TemporaryDirectory() as tmp:
        lib = shutil.copy(_hip.lib_path(), os.path.join(tmp, "lib.so"))       # --offloading writes next to its input
        subprocess.run([objdump, "--offloading", lib], check=True, capture_output=True, cwd=tmp)
        bundles = [f for f in os.listdir(tmp) if f.endswith("gfx950")]
        assert len(bundles) >= 9, bundles
        packed, bad, kernels = 0, [], 0
        for f in bundles:
            dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = ""
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    cur = m.group(1)
                    kernels += 1
                    continue
                if re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
                    packed += 1
                    if re.search(r"\bop_sel:\[", line):
                        bad.append((cur[:80], line.split("//")[0].strip()))
    assert kernels > 100, kernels                     # the disassembly really covered the library
    assert packed >= 100, packed                      # ... down to the instruction level (conv3d_wres's hand-written pairs)
    assert not bad, bad[:5]


def test_attention_dropout_rate_is_quantised_and_small_rates_warn():
    """csrc/attention.hip quantises the attention-probability dropout rate to 1/256; ops documents the effective
    rate and warns when a requested rate rounds to none (host logic: the warning fires before any kernel is called)"""
    from multimodal_eeg_fmri_amd import ops
    assert ops.attn_effective_dropout(0.1) == 26 / 256 and ops.attn_effective_dropout(0.3) == 77 / 256
    assert ops.attn_effective_dropout(0.001) == 0.0 and ops.attn_effective_dropout(0.0) == 0.0
    with pytest.warns(UserWarning, match="below the kernels' resolution"):
        with pytest.raises(Exception):                       # (a CPU tensor: the HIP path refuses it right after the warning)
            ops.attention(torch.zeros(1, 4, 96), 4, False, drop_p=0.001)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """under a launcher whose WORLD_SIZE differs from --gpus the bench stops with a message before any GPU call (no assert
    trace, no hang)"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "launches its own ranks" in r.stderr


def test_bench_self_launch_relays_a_failing_rank():
    """`python bench.py --gpus 2` without a launcher becomes the parent of its own two ranks (torch.distributed.run) and
    exits NON-ZERO when they fail - which they must here, where there is no GPU - without printing a result line.  The
    parent itself never imports torch: it cannot have touched the GPU."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    probe = ("import runpy, sys; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']\n"
             "import subprocess\n"
             "real = subprocess.Popen\n"
             "def spy(cmd, **kw):\n"
             "    assert 'torch' not in sys.modules, 'the parent imported torch before launching its ranks'\n"
             "    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '2'\n"
             "    assert '--master-addr' in cmd and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'\n"
             "    return real(cmd, **kw)\n"
             "subprocess.Popen = spy\n"
             f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: the ranks would run")
    assert r.returncode != 0, r.stdout[-800:] + r.stderr[-800:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "AssertionError" not in r.stderr, r.stderr[-1500:]

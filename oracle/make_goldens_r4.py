#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Round-4 addition to tests/golden/ (same rules as make_goldens.py: the REFERENCE is
imported read-only from /root/reference in the build container; only small arrays are stored, no reference text):

  f3_lite_datasets.npz   EEGDatasetERP / EEGDatasetPW / EEGDatasetCONN + aggregate_features
                         (EEG_CODE/run_training_lite.py:62-259) on a synthetic MATLAB-v5 tree: every item
                         tuple of the three datasets and the per-subject aggregates, from the reference's own classes.
                         The product's classes are checked bit-equal here and again by the CPU test, which rebuilds
                         the tree from the fixture.

h5py is not installed in the image: the reference module's `import h5py` gets an EMPTY module object, every h5py.* use
raises and the reference takes its own scipy.io.loadmat branch - its HDF5 (MATLAB v7.3) branch stays UNPINNED.

Run:  python oracle/make_goldens_r4.py
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def lite_tree(rng):
    """{relative path: {variable: array}} - names chosen to hit every branch of the three loaders"""
    r = lambda *s: rng.standard_normal(s)        # noqa: E731
    return {
        # CONN: a matrix -> strict upper triangle; a 3-D array -> flattened; an unknown variable name -> first non-dunder;
        # a name only the SECOND glob pattern matches (no "sub" prefix)
        "conn/conn_sub001_alpha_open.mat": {"conn": r(6, 6)},
        "conn/conn_sub001_alpha_close.mat": {"connectivity": r(6, 6)},
        "conn/conn_sub002_alpha_open.mat": {"zz_other": r(6, 6)},
        "conn/conn_sub002_beta_open.mat": {"data": r(1, 3, 5)},          # 15 values, as a 6 x 6 upper triangle
        "conn/c_003_alpha.mat": {"conn": r(6, 6)},
        # PW: (channels, samples) arrays kept 2-D; "1_Hz" also matches "11_Hz" (glob substring, as in the reference)
        "pw/pw_sub001_alpha_1_Hz.mat": {"powspctrm": r(4, 10)},
        "pw/pw_sub001_alpha_11_Hz.mat": {"pw": r(4, 10)},
        "pw/pw_sub002_alpha_1_Hz.mat": {"power": r(4, 10)},
        "pw/pw_sub002_beta_2_Hz.mat": {"anything": r(4, 10)},
        "pw/pw_sub003_alpha_2_Hz.mat": {"data": r(4, 10)},
        # ERP
        "erp/erp_sub001_alpha_1_Hz.mat": {"ERP": r(4, 16)},
        "erp/erp_sub001_beta_2_Hz.mat": {"erp": r(4, 16)},
        "erp/erp_sub002_alpha_1_Hz.mat": {"data": r(4, 16)},
        "erp/erp_sub003_alpha_2_Hz.mat": {"first": r(4, 16)},
        "erp/broken_sub003_alpha_1_Hz.mat": None,             # not a MATLAB file: skipped by both
    }


def write_tree(tmp, tree):
    from scipy.io import savemat
    for rel, content in tree.items():
        path = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if content is None:
            with open(path, "wb") as fh:
                fh.write(b"not a mat file")
        else:
            savemat(path, content)


ARGS = dict(subjects=[1, 2, 3, 4], bands={"alpha": "Alpha", "beta": "Beta"}, freqs=["1_Hz", "2_Hz"], conds=["open", "close"],
            labels={1: 0, 2: 1, 3: 1})


def run_datasets(mod, tmp):
    """the three datasets + their aggregates from module ``mod`` (the reference's or the product's)"""
    a = ARGS
    with contextlib.redirect_stdout(io.StringIO()):
        ds = {"erp": mod.EEGDatasetERP(a["subjects"], a["bands"], a["freqs"], os.path.join(tmp, "erp"), labels=a["labels"]),
              "pw": mod.EEGDatasetPW(a["subjects"], a["bands"], a["freqs"], os.path.join(tmp, "pw"), labels=a["labels"]),
              "conn": mod.EEGDatasetCONN(a["subjects"], a["bands"], a["conds"], os.path.join(tmp, "conn"), labels=a["labels"])}
        agg = {k: mod.aggregate_features(d, k) for k, d in ds.items()}
    return ds, agg


def flatten(ds, agg):
    """-> {name: array} in a canonical order (files sorted by their metadata: glob order is the file system's)"""
    out = {}
    for k, d in ds.items():
        items = sorted((d[i] for i in range(len(d))), key=lambda t: (t[1], t[2], t[3], tuple(t[0].shape), float(t[0].flatten()[0])))
        out[f"{k}_n"] = np.array(len(items))
        for i, (feat, subj, band, third, y) in enumerate(items):
            out[f"{k}_{i}_feat"] = feat.numpy()
            out[f"{k}_{i}_meta"] = np.array([str(subj), band, third, str(y)])
        feats, labels = agg[k]
        out[f"{k}_agg_subjects"] = np.array(sorted(feats))
        for s in sorted(feats):
            out[f"{k}_agg_{s}"] = feats[s].numpy()
        out[f"{k}_agg_labels"] = np.array([labels[s] for s in sorted(labels)])
    return out


def main():
    if "h5py" not in sys.modules:
        sys.modules["h5py"] = types.ModuleType("h5py")
    sys.path.insert(0, os.path.join(REF, "EEG_CODE"))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as scratch:
        os.chdir(scratch)                                  # (nothing may be written next to the reference)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                import run_training_lite as ref_rtl        # noqa: E402  the REFERENCE's module (EEG_CODE on sys.path)
        finally:
            os.chdir(cwd)
    assert ref_rtl.__file__.startswith(REF), ref_rtl.__file__
    import multimodal_eeg_fmri_amd.run_training_lite as our_rtl
    rng = np.random.default_rng(11)
    tree = lite_tree(rng)
    with tempfile.TemporaryDirectory() as tmp:
        write_tree(tmp, tree)
        want = flatten(*run_datasets(ref_rtl, tmp))
        got = flatten(*run_datasets(our_rtl, tmp))
    assert list(want) == list(got), (list(want), list(got))
    for k in want:
        assert want[k].shape == got[k].shape and want[k].dtype == got[k].dtype and np.array_equal(want[k], got[k]), k
    # (the CONN fallback pattern re-matches files of the same subject and band: duplicates, as in the reference)
    assert (int(want["erp_n"]), int(want["pw_n"]), int(want["conn_n"])) == (4, 5, 8), [int(want[k]) for k in ("erp_n", "pw_n", "conn_n")]
    fx = {"paths": np.array(list(tree)), "expected_keys": np.array(list(want))}
    for i, content in enumerate(tree.values()):
        if content is None:
            fx[f"file_{i}_var"] = np.array("")
        else:
            (var, arr), = content.items()
            fx[f"file_{i}_var"] = np.array(var)
            fx[f"file_{i}_arr"] = arr
    for i, v in enumerate(want.values()):
        fx[f"exp_{i}"] = v
    np.savez_compressed(os.path.join(OUT, "f3_lite_datasets.npz"), **fx)
    print(f"f3_lite_datasets.npz: {len(want)} arrays from the reference's EEGDataset* classes; product classes bit-equal "
          "(reference HDF5 branch unpinned: h5py absent)")


if __name__ == "__main__":
    main()

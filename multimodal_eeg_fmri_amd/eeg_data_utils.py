"""EEG feature / label loaders of the bridge pipeline: same functions, arguments,
file-name patterns and dictionary keys as the reference's ``EEG_CODE/eeg_data_utils.py``.  Pure host
I/O feeding the HIP path; ``h5py`` (MATLAB v7.3 ERP files) is optional - without it ERP files are
read through ``scipy.io.loadmat``, which is also the reference's fallback branch (:168-180).
"""
from __future__ import annotations

import glob
import logging
import os
from pathlib import Path

import numpy as np

logger = logging.getLogger(__name__)


def load_eeg_labels(label_dir, binary=True):
    """{subject: label} from ``medical_score.csv`` (reference :19-44): rows without a
    'Postoperative evaluation' are dropped, 'subNN' ids become ints, score <= 2 -> 0, else 1 when
    ``binary`` (else the raw score)."""
    import pandas as pd
    csv_path = os.path.join(str(label_dir), "medical_score.csv")
    if not os.path.exists(csv_path):
        raise FileNotFoundError(f"Label file not found: {csv_path}")
    df = pd.read_csv(csv_path).dropna(subset=["Postoperative evaluation"])
    ids = df["Subject"]
    ids = ids.str.replace("sub", "", regex=False).astype(int) if ids.dtype == object else ids.astype(int)
    labels = {}
    for subj, score in zip(ids, df["Postoperative evaluation"]):
        labels[int(subj)] = 0 if score <= 2 else 1 if binary else score
    return labels


def _first_matrix(path, flatten):
    """first non-dunder variable of a MATLAB v5 file as float32, NaN -> 0"""
    from scipy.io import loadmat
    mat = loadmat(path)
    for key in mat:
        if not key.startswith("_"):
            data = np.array(mat[key], dtype=np.float32)
            return np.nan_to_num(data.flatten() if flatten else data, nan=0.0)
    return None


def load_eeg_conn_features(conn_dir, subject_list, band_list, cond_list):
    """{(subject, band_key, condition, 0): flat float32} from ``conn_<BandName>_<cond>_subNN.mat``,
    falling back to ``conn_<band_key>_...`` (reference :47-84); ``band_list`` maps key -> name."""
    out = {}
    for subj in subject_list:
        for band_key, band_name in band_list.items():
            for cond in cond_list:
                files = sorted(glob.glob(str(Path(conn_dir) / f"conn_{band_name}_{cond}_sub{subj:02d}.mat")))
                if not files:
                    files = sorted(glob.glob(str(Path(conn_dir) / f"conn_{band_key}_{cond}_sub{subj:02d}.mat")))
                for f in files:
                    try:
                        data = _first_matrix(f, flatten=True)
                        if data is not None:
                            out[(subj, band_key, cond, 0)] = data
                    except Exception as e:
                        logger.warning("Error loading %s: %s", f, e)
    logger.info("Loaded %d EEG connectivity samples", len(out))
    return out


def load_eeg_pw_features(pw_dir, subject_list, band_list, freq_list):
    """{(subject, band, freq, 0): flat float32} from ``powspctrm_<band>_<freq>_subNN.mat`` (:87-119)."""
    out = {}
    for subj in subject_list:
        for band in band_list:
            for freq in freq_list:
                for f in sorted(glob.glob(str(Path(pw_dir) / f"powspctrm_{band}_{freq}_sub{subj:02d}.mat"))):
                    try:
                        data = _first_matrix(f, flatten=True)
                        if data is not None:
                            out[(subj, band, freq, 0)] = data
                    except Exception as e:
                        logger.warning("Error loading %s: %s", f, e)
    logger.info("Loaded %d EEG power spectrum samples", len(out))
    return out


def _erp_from_hdf5(path):
    """MATLAB v7.3 layout (:140-165): group erp_struct | erp | first key; dataset avg, else trial
    (3-D -> mean over trials), else the first dataset with >= 2 dims."""
    import h5py
    with h5py.File(path, "r") as hf:
        group = hf["erp_struct"] if "erp_struct" in hf else hf["erp"] if "erp" in hf else hf[list(hf.keys())[0]]
        if "avg" in group:
            return np.array(group["avg"], dtype=np.float32)
        if "trial" in group:
            data = np.array(group["trial"], dtype=np.float32)
            return data.mean(axis=0) if data.ndim == 3 else data
        for key in group.keys():
            cand = group[key]
            if hasattr(cand, "shape") and len(cand.shape) >= 2:
                return np.array(cand, dtype=np.float32)
    return None


def load_eeg_erp_features(erp_dir, subject_list, band_list, freq_list):
    """{(subject, band, freq, 0): 2-D float32} from ``ERP_subNN_<band>_<freq>*.mat`` (:122-185): HDF5
    first, MATLAB-v5 ``loadmat`` (array kept 2-D) when that fails or h5py is not installed."""
    out = {}
    for subj in subject_list:
        for band in band_list:
            for freq in freq_list:
                for f in sorted(glob.glob(str(Path(erp_dir) / f"ERP_sub{subj:02d}_{band}_{freq}*.mat"))):
                    try:
                        data = _erp_from_hdf5(f)
                        if data is not None:
                            out[(subj, band, freq, 0)] = np.nan_to_num(data, nan=0.0)
                    except Exception as e:
                        try:
                            data = _first_matrix(f, flatten=False)
                            if data is not None:
                                out[(subj, band, freq, 0)] = data
                        except Exception:
                            logger.warning("Error loading ERP %s: %s", f, e)
    logger.info("Loaded %d EEG ERP samples", len(out))
    return out

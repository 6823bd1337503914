"""V4-Lite tri-modal training entry point on the MI355X HIP path.

Counterpart of the reference's ``EEG_CODE/run_training_lite.py`` with the same
call surface (``main``, ``ImprovedTriModalFusionNetLite``, ``collate_balanced``,
``aggregate_features``, ``load_labels``, ``normalize_modality``,
``vec_upper_triangle``) and the same protocol: StratifiedGroupKFold (:431-433),
LabelSmoothingCrossEntropy(0.1) + AdamW(wd 0.01) + CosineAnnealingWarmup(3) +
EarlyStopping(15) (:465-468), grad-clip 1.0 (:487), best-F1 state restore
(:510-520).  The reference's ``main()`` cannot run against its own ``Config``
 and reads private clinical ``.mat`` files through
``EEGDatasetERP / EEGDatasetPW / EEGDatasetCONN`` (:62-231, :367-386); those three
dataset classes are here too (same constructor arguments, glob patterns, variable
names searched, per-file z-score and item tuples), and ``main()`` takes such a tree
when the configured directories hold one (``source="disk"`` / ``"auto"``), else
drives the same loop on the synthetic stand-in described by ``Config.synthetic``
(BASELINE config #1: 8 ch x 256 samples, conn 459).
"""
from __future__ import annotations

import glob
from collections import defaultdict
from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .config import Config, set_seed
from .optim import FusedAdamW
from .crossmodal_v4_enhancements import (CosineAnnealingWarmup, EarlyStopping,
                                         EnhancedTriModalFusionNetV4Lite,
                                         LabelSmoothingCrossEntropy, get_lite_fusion_weights)


# ----------------------------------------------------------------- helpers
def normalize_modality(feat, eps=1e-8):
    """global z-score of one modality tensor (run_training_lite.py:48-51)."""
    return (feat - feat.mean()) / (feat.std() + eps)


def vec_upper_triangle(mat):
    """strict upper triangle of a square matrix as a vector (:53-56)."""
    return mat[np.triu_indices(mat.shape[0], k=1)]


def aggregate_features(dataset, name="data"):
    """mean of every subject's samples -> ({subject: tensor}, {subject: label})."""
    per_subject = defaultdict(list)
    labels = {}
    for i in range(len(dataset)):
        sample = dataset[i]
        feat = sample[0].numpy() if isinstance(sample[0], torch.Tensor) else np.asarray(sample[0])
        per_subject[sample[1]].append(feat)
        labels[sample[1]] = sample[-1]
    agg = {s: torch.tensor(np.mean(np.stack(v, axis=0), axis=0), dtype=torch.float32)
           for s, v in per_subject.items()}
    print(f"Aggregated {len(dataset)} {name} samples to {len(agg)} subjects")
    return agg, labels


def load_labels(label_path, binary=True):
    """CSV -> {subject: label}; column sniffing as in the reference (:261-295)."""
    import pandas as pd
    df = pd.read_csv(label_path)
    subj_col = next((c for c in ("subject", "Subject", "subj", "ID", "id", "SubjectID") if c in df.columns),
                    df.columns[0])
    label_col = next((c for c in ("label", "Label", "class", "Class", "score", "Score", "y", "target")
                      if c in df.columns), df.columns[1])
    labels = {}
    for _, row in df.iterrows():
        lab = int(row[label_col])
        labels[int(row[subj_col])] = (0 if lab <= 1 else 1) if binary else lab
    print(f"Loaded {len(labels)} labels")
    return labels


def collate_balanced(batch):
    """dict- or tuple-style samples -> (erp, pw, conn, labels, subjects)."""
    cols = ([], [], [], [], [])
    keys = ("erp", "pw", "conn", "label", "subject")
    for sample in batch:
        vals = [sample[k] for k in keys] if isinstance(sample, dict) else list(sample[:5])
        for c, v in zip(cols, vals):
            c.append(v)
    return (torch.stack(cols[0]), torch.stack(cols[1]), torch.stack(cols[2]),
            torch.tensor(cols[3], dtype=torch.long), cols[4])


# ------------------------------------------------ on-disk datasets (:62-231)
def _mat_variable(path, keys, transpose_hdf5: bool):
    """the array one feature file holds: MATLAB v7.3 (HDF5) first - variable = first of ``keys`` present, else the
    file's first; column-major on disk, hence the transpose the reference applies to power / ERP files (:131, :187) but
    not to connectivity matrices (:86) - and MATLAB v5 through scipy.io.loadmat when that fails (h5py not installed, or
    not an HDF5 file), where the fallback variable is the first non-dunder one."""
    try:
        import h5py
        with h5py.File(path, "r") as f:
            key = next((k for k in keys if k in f), None) or list(f.keys())[0]
            data = np.array(f[key])
            return data.T if transpose_hdf5 else data
    except Exception:  # noqa: BLE001 - the reference's bare except: anything wrong with the HDF5 route -> loadmat
        import scipy.io
        mat = scipy.io.loadmat(path)
        key = next((k for k in keys if k in mat), None) or [k for k in mat if not k.startswith("_")][0]
        return mat[key]


class _EEGMatDataset(torch.utils.data.Dataset):
    """shared body of the three per-file datasets: one sample per matching ``.mat`` file,
    ``(feature tensor, subject, band, third key, label)``; a file that cannot be read is skipped (reported when
    ``verbose``)."""
    KEYS = ()
    NAME = ""
    TRANSPOSE_HDF5 = True

    def _patterns(self, root, subj_str, band, third):
        return [str(root / f"*sub{subj_str}*{band}*{third}*.mat")]

    def _feature(self, data):
        return normalize_modality(data.astype(np.float32))

    def __init__(self, subj_list, band_list, third_list, root, labels=None, verbose=False):
        self.samples = []
        root = Path(root)
        for subj in subj_list:
            for band in band_list.keys():
                for third in third_list:
                    files = []
                    for pat in self._patterns(root, f"{subj:03d}", band, third):
                        files = glob.glob(pat)
                        if files:
                            break
                    for fpath in files:
                        try:
                            feat = self._feature(_mat_variable(fpath, self.KEYS, self.TRANSPOSE_HDF5))
                            self.samples.append((feat, subj, band, third, labels.get(subj, 0) if labels else 0))
                        except Exception as e:  # noqa: BLE001
                            if verbose:
                                print(f"Failed to load {fpath}: {e}")
        print(f"  Loaded {len(self.samples)} {self.NAME} samples")

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        feat, subj, band, third, y = self.samples[idx]
        return torch.tensor(feat, dtype=torch.float32), subj, band, third, y


class EEGDatasetCONN(_EEGMatDataset):
    """``EEGDatasetCONN(subj_list, band_list, cond_list, conn_dir, labels=None, verbose=False)`` (:62-126): files
    ``*sub<NNN>*<band>*<cond>*.mat``, else ``*<NNN>*<band>*.mat``; variable conn | connectivity | data; a matrix
    becomes its strict upper triangle, anything else is flattened; z-scored; float32."""
    KEYS = ("conn", "connectivity", "data")
    NAME = "CONN"
    TRANSPOSE_HDF5 = False

    def _patterns(self, root, subj_str, band, third):
        return [str(root / f"*sub{subj_str}*{band}*{third}*.mat"), str(root / f"*{subj_str}*{band}*.mat")]

    def _feature(self, data):
        flat = vec_upper_triangle(data) if data.ndim == 2 else data.flatten()
        return normalize_modality(flat).astype(np.float32)


class EEGDatasetPW(_EEGMatDataset):
    """``EEGDatasetPW(subj_list, band_list, freq_list, pw_dir, labels=None, verbose=False)`` (:129-179): files
    ``*sub<NNN>*<band>*<freq>*.mat``; variable powspctrm | pw | power | data; float32, z-scored, shape kept."""
    KEYS = ("powspctrm", "pw", "power", "data")
    NAME = "PW"


class EEGDatasetERP(_EEGMatDataset):
    """``EEGDatasetERP(subj_list, band_list, freq_list, erp_dir, labels=None, verbose=False)`` (:182-231): as the power
    dataset with variable ERP | erp | data."""
    KEYS = ("ERP", "erp", "data")
    NAME = "ERP"


class AggregatedTriModalDataset(torch.utils.data.Dataset):
    """one (erp, pw, conn) sample per subject present in all three aggregated dictionaries and in the labels.  The
    reference's main() hands its aggregates to BalancedTriModalDataset (:394-396), which FLATTENS every modality
    (crossmodal_v4_enhancements.py:1020-1058) and so could never feed the Conv1d encoders it then builds (one of the
    reasons that main() does not run); here the (channels, samples) arrays stay 2-D."""

    def __init__(self, erp_agg: Dict, pw_agg: Dict, conn_agg: Dict, label_dict: Dict):
        common = sorted(set(erp_agg) & set(pw_agg) & set(conn_agg) & set(label_dict))
        self.samples = [{"erp": torch.as_tensor(erp_agg[s], dtype=torch.float32), "pw": torch.as_tensor(pw_agg[s], dtype=torch.float32),
                         "conn": torch.as_tensor(conn_agg[s], dtype=torch.float32).flatten(),
                         "label": int(label_dict[s]), "subject": s} for s in common]
        print(f"AggregatedTriModalDataset: {len(self.samples)} subjects present in all three modalities")

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        return self.samples[i]


def _disk_labels(label_path, binary=True):
    """``config.label_path``: a CSV for load_labels (what the reference's main() passes, :363), or - as Config's default
    is a DIRECTORY (config.py:33) - the directory holding eeg_data_utils' ``medical_score.csv``"""
    label_path = Path(label_path)
    if label_path.is_dir():
        from .eeg_data_utils import load_eeg_labels
        return load_eeg_labels(label_path, binary=binary)
    return load_labels(label_path, binary=binary)


def load_disk_dataset(config: Config):
    """the reference's loading steps (:360-396) on ``config``'s directories -> AggregatedTriModalDataset, or None when
    a modality has no readable file"""
    label_dict = _disk_labels(config.label_path, binary=True)
    erp = EEGDatasetERP(config.subject_list, config.bands, config.freq_bands, config.eeg_path_erp, labels=label_dict)
    pw = EEGDatasetPW(config.subject_list, config.bands, config.freq_bands, config.eeg_path_pw, labels=label_dict)
    conn = EEGDatasetCONN(config.subject_list, config.bands, config.func_segments, config.eeg_path_conn, labels=label_dict)
    if len(erp) == 0 or len(pw) == 0 or len(conn) == 0:
        print("ERROR: No data loaded!")
        return None
    erp_agg, _ = aggregate_features(erp, "ERP")
    pw_agg, _ = aggregate_features(pw, "PW")
    conn_agg, _ = aggregate_features(conn, "CONN")
    return AggregatedTriModalDataset(erp_agg, pw_agg, conn_agg, label_dict)


class SyntheticTriModalDataset(torch.utils.data.Dataset):
    """stand-in for BalancedTriModalDataset (crossmodal_v4_enhancements.py:955-1077):
    one aggregated (erp, pw, conn) sample per subject with a class-dependent shift."""

    def __init__(self, spec: Dict, n_classes: int = 2):
        g = torch.Generator().manual_seed(int(spec["seed"]))
        n = int(spec["subjects"])
        self.samples = []
        for s in range(n):
            y = s % n_classes
            shift = 0.35 * (2 * y - 1)
            self.samples.append({
                "erp": torch.randn(spec["erp_channels"], spec["samples"], generator=g) + shift,
                "pw": torch.randn(spec["pw_channels"], spec["samples"], generator=g) - shift,
                "conn": torch.randn(spec["conn_features"], generator=g) + 0.5 * shift,
                "label": y, "subject": s + 1})

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        return self.samples[i]


# ------------------------------------------------------------ model wrapper
class ImprovedTriModalFusionNetLite(nn.Module):
    """argument order (pw, erp, conn) as in the reference wrapper (:302-328)."""

    def __init__(self, in_pw_dim, in_erp_dim, in_conn_dim, fusion_dim=96, num_classes=2,
                 dropout=0.4, conn_boost=1.3):
        super().__init__()
        self.model = EnhancedTriModalFusionNetV4Lite(
            erp_channels=in_erp_dim, pw_channels=in_pw_dim, conn_features=in_conn_dim,
            hidden_dim=fusion_dim, num_classes=num_classes, dropout=dropout, conn_boost=conn_boost)
        self.fusion_weight_history = []

    def forward(self, pw, erp, conn):
        logits, _ = self.model(erp, pw, conn, return_fusion_weights=True)
        return logits

    def get_fusion_weights(self):
        return get_lite_fusion_weights(self.model)

    def track_fusion_weights(self):
        w = self.get_fusion_weights()
        if w:
            self.fusion_weight_history.append(w)


# --------------------------------------------------------------------- main
def _evaluate(model, loader, device):
    model.eval()
    preds, targets = [], []
    with torch.no_grad():
        for erp, pw, conn, y, _ in loader:
            logits = model(pw.to(device), erp.to(device), conn.to(device))
            preds.extend(logits.argmax(dim=1).cpu().tolist())
            targets.extend(y.tolist())
    return np.array(preds), np.array(targets)


def main(config: Optional[Config] = None, max_epochs: Optional[int] = None, source: str = "auto"):
    """5-fold cross-validated training of the Lite tri-modal net.  ``source``: "disk" = the ``.mat`` tree under
    ``config``'s directories (the reference's own route, :360-396), "synthetic" = ``Config.synthetic``, "auto" = disk
    when all three feature directories and the label path exist, else synthetic."""
    from sklearn.metrics import accuracy_score, f1_score
    from sklearn.model_selection import StratifiedGroupKFold
    from torch.utils.data import DataLoader, Subset

    if not torch.cuda.is_available():
        raise RuntimeError("run_training_lite.main(): the HIP path needs an MI355X (no CPU fallback)")
    device = torch.device("cuda")
    set_seed(42)
    config = config or Config(None)
    on_disk = all(Path(p).exists() for p in (config.eeg_path_erp, config.eeg_path_pw, config.eeg_path_conn, config.label_path))
    if source == "disk" or (source == "auto" and on_disk):
        dataset = load_disk_dataset(config)
        if dataset is None or len(dataset) == 0:
            print("ERROR: No samples in balanced dataset!")
            return None
    else:
        dataset = SyntheticTriModalDataset(config.synthetic)
    labels = np.array([s["label"] for s in dataset.samples])
    subjects = np.array([s["subject"] for s in dataset.samples])
    n_classes = len(np.unique(labels))
    erp_ch, pw_ch = dataset[0]["erp"].shape[0], dataset[0]["pw"].shape[0]
    conn_dim = dataset[0]["conn"].numel()
    epochs = max_epochs or config.epochs
    n_splits = min(config.n_splits, len(np.unique(subjects)))
    splits = StratifiedGroupKFold(n_splits=n_splits, shuffle=True, random_state=42).split(
        np.zeros(len(dataset)), labels, groups=subjects)
    results = []
    for fold, (tr, te) in enumerate(splits, 1):
        train_loader = DataLoader(Subset(dataset, tr), batch_size=config.batch_size, shuffle=True,
                                  collate_fn=collate_balanced, drop_last=len(tr) % config.batch_size == 1)
        test_loader = DataLoader(Subset(dataset, te), batch_size=config.batch_size, collate_fn=collate_balanced)
        model = ImprovedTriModalFusionNetLite(pw_ch, erp_ch, conn_dim, fusion_dim=96, num_classes=n_classes,
                                              dropout=0.4, conn_boost=1.3).to(device)
        criterion = LabelSmoothingCrossEntropy(smoothing=0.1)
        # clip_grad_norm_(1.0) + AdamW(weight_decay=0.01) (reference :466,:487-488) as one fused kernel pair
        optimizer = FusedAdamW(model.parameters(), lr=config.learning_rate, weight_decay=0.01, max_grad_norm=1.0)
        scheduler = CosineAnnealingWarmup(optimizer, warmup_epochs=3, total_epochs=epochs)
        stopper = EarlyStopping(patience=15, mode="max")
        best_f1, best_state = 0.0, None
        for epoch in range(1, epochs + 1):
            model.train()
            for erp, pw, conn, y, _ in train_loader:
                optimizer.zero_grad()
                loss = criterion(model(pw.to(device), erp.to(device), conn.to(device)), y.to(device))
                loss.backward()
                optimizer.step()
            scheduler.step()
            preds, targets = _evaluate(model, test_loader, device)
            f1 = f1_score(targets, preds, average="weighted")
            if f1 > best_f1:
                best_f1 = f1
                best_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            if stopper(f1):
                break
        if best_state:
            model.load_state_dict({k: v.to(device) for k, v in best_state.items()})
        preds, targets = _evaluate(model, test_loader, device)
        results.append({"Accuracy": accuracy_score(targets, preds),
                        "F1": f1_score(targets, preds, average="weighted")})
        print(f"FOLD {fold}: Acc={results[-1]['Accuracy']:.4f} F1={results[-1]['F1']:.4f}")
    print(f"Accuracy: {np.mean([r['Accuracy'] for r in results]):.4f} "
          f"+/- {np.std([r['Accuracy'] for r in results]):.4f}")
    return results


if __name__ == "__main__":
    main()

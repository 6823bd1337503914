"""fMRI encoders on the MI355X HIP path.

* ``ActivationEncoder`` / ``ConnectivityEncoder`` / ``fMRIFusionNet`` mirror the
  reference's tabular model (``fMRI_CODE/fmri_utils.py:23-108`` ==
  ``fMRI_CODE/run_fmri_v11.py:272-424``): same names, signatures, state_dict.
* ``fMRIVolumeEncoder3D`` is the north-star 3-D voxel encoder.  The reference
  has **no** volume code, so this class is an extension defined
  here in the reference's Conv->BN->GELU->Pool idiom and ending in the 64-d
  feature the bridge expects (``bridge_utils.py:28``); parity is unpinned by the
  reference and checked against ``torch.nn.Conv3d`` semantics instead.
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn

from . import ops

logger = logging.getLogger(__name__)


def _mlp(in_dim, hidden_dim, dropout):
    return nn.Sequential(
        nn.Linear(in_dim, hidden_dim * 2), nn.BatchNorm1d(hidden_dim * 2), nn.ReLU(), nn.Dropout(dropout),
        nn.Linear(hidden_dim * 2, hidden_dim), nn.BatchNorm1d(hidden_dim), nn.ReLU(), nn.Dropout(dropout))


class ActivationEncoder(nn.Module):
    def __init__(self, in_dim: int, hidden_dim: int = 64, dropout: float = 0.3):
        super().__init__()
        self.encoder = _mlp(in_dim, hidden_dim, dropout)
        self.drop_p = dropout

    def forward(self, x):
        return ops.fmri_mlp_forward(self.encoder, x, self.drop_p, self.training)


class ConnectivityEncoder(ActivationEncoder):
    pass


class fMRIFusionNet(nn.Module):
    """softmax-weighted concat of the two MLP features -> Linear-BN-ReLU -> head."""

    def __init__(self, activation_dim: int, connectivity_dim: int, hidden_dim: int = 64,
                 num_classes: int = 2, dropout: float = 0.4, task: str = "classification"):
        super().__init__()
        self.task = task
        self.activation_encoder = ActivationEncoder(activation_dim, hidden_dim, dropout)
        self.connectivity_encoder = ConnectivityEncoder(connectivity_dim, hidden_dim, dropout)
        self.fusion = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.BatchNorm1d(hidden_dim),
                                    nn.ReLU(), nn.Dropout(dropout))
        self.activation_weight = nn.Parameter(torch.ones(1) * 0.5)
        self.connectivity_weight = nn.Parameter(torch.ones(1) * 0.5)
        out_dim = num_classes if task == "classification" else 1
        self.head = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(),
                                  nn.Dropout(dropout), nn.Linear(hidden_dim // 2, out_dim))
        self.drop_p = dropout

    def forward(self, activation, connectivity, return_features: bool = False):
        output, fused = ops.fmri_fusion_forward(self, activation, connectivity)
        if self.task == "regression":
            output = output.squeeze(-1)
        return (output, fused) if return_features else output

    def get_fusion_weights(self):
        with torch.no_grad():
            w = torch.softmax(torch.stack([self.activation_weight, self.connectivity_weight]), dim=0)
        return {"activation": w[0].item(), "connectivity": w[1].item()}


class _fMRISingleBranch(nn.Module):
    """one tabular encoder + the Linear-ReLU-Dropout-Linear head (``run_fmri_v11.py:311-370``); state_dict keys
    ``encoder.encoder.{0,1,4,5}.*``, ``head.{0,3}.*`` as in the reference."""
    _encoder_cls = ActivationEncoder

    def __init__(self, in_dim: int, hidden_dim: int = 64, num_classes: int = 2, dropout: float = 0.4,
                 task: str = "classification"):
        super().__init__()
        self.task = task
        self.encoder = self._encoder_cls(in_dim, hidden_dim, dropout)
        self.head = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(), nn.Dropout(dropout),
                                  nn.Linear(hidden_dim // 2, num_classes if task == "classification" else 1))
        self.drop_p = dropout

    def _run(self, x):
        output = ops.fmri_single_forward(self, x)
        return output.squeeze(-1) if self.task == "regression" else output


class fMRIActivationOnly(_fMRISingleBranch):
    """activation features only; ``forward(activation, connectivity=None)`` ignores the second argument
    (``run_fmri_v11.py:311-337``), so the three model kinds share ``train_epoch`` / ``evaluate``."""

    def forward(self, activation, connectivity=None):
        return self._run(activation)


class fMRIConnectivityOnly(_fMRISingleBranch):
    """connectivity features only (``run_fmri_v11.py:340-366``)."""
    _encoder_cls = ConnectivityEncoder

    def forward(self, activation=None, connectivity=None):
        return self._run(connectivity)


class fMRIVolumeEncoder3D(nn.Module):
    """(B, 1, D, H, W) fp32 volume -> (B, out_dim) feature.

    Conv3d(1->w0,k3,p1)-BN-GELU-MaxPool(2) / Conv3d(w0->w1)-BN-GELU-MaxPool(2) /
    Conv3d(w1->w2)-BN-GELU / global average pool / Linear(w2->out_dim)-GELU.
    Layer 1 (K=27) is an HBM-bound direct conv; layers 2-3 are LDS-staged
    implicit-GEMM tiles feeding bf16 MFMA with fp32 accumulation.
    """

    def __init__(self, in_channels: int = 1, out_dim: int = 64,
                 widths=(32, 64, 128), dropout: float = 0.3):
        super().__init__()
        w0, w1, w2 = widths
        self.conv_layers = nn.Sequential(
            nn.Conv3d(in_channels, w0, kernel_size=3, padding=1), nn.BatchNorm3d(w0),
            nn.GELU(), nn.MaxPool3d(2), nn.Dropout(dropout),
            nn.Conv3d(w0, w1, kernel_size=3, padding=1), nn.BatchNorm3d(w1),
            nn.GELU(), nn.MaxPool3d(2), nn.Dropout(dropout),
            nn.Conv3d(w1, w2, kernel_size=3, padding=1), nn.BatchNorm3d(w2),
            nn.GELU(), nn.Dropout(dropout))
        self.output_proj = nn.Sequential(nn.AdaptiveAvgPool3d(1), nn.Flatten(),
                                         nn.Linear(w2, out_dim), nn.GELU(), nn.Dropout(dropout))
        self.drop_p = dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.volume_encoder_forward(self, x)


# ---------------------------------------------------------------------------
# host side of the tabular fMRI pipeline: dataset, collate, the epoch
# loops and the CSV loaders.  Same names, arguments and return values as the reference; the model
# calls inside the loops run on the HIP kernels.
# ---------------------------------------------------------------------------

class fMRIDataset(torch.utils.data.Dataset):
    """subjects present in all three dicts, ascending (``run_fmri_v11.py:216-257``); item =
    ``(activation, connectivity, class_label, reg_label, subject)``, ``reg_label`` 0.0 when absent."""

    def __init__(self, activation_features, connectivity_features, class_labels, reg_labels=None, transform=None):
        self.transform = transform
        common = set(activation_features) & set(connectivity_features) & set(class_labels)
        self.samples = [{"activation": activation_features[s], "connectivity": connectivity_features[s],
                         "class_label": class_labels[s],
                         "reg_label": reg_labels[s] if reg_labels and s in reg_labels else 0.0,
                         "subject": s} for s in sorted(common)]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        s = self.samples[idx]
        act, conn = s["activation"], s["connectivity"]
        if self.transform:
            act, conn = self.transform(act), self.transform(conn)
        return act, conn, s["class_label"], s["reg_label"], s["subject"]


def collate_fmri(batch):
    """(``run_fmri_v11.py:259-266``)"""
    return (torch.stack([b[0] for b in batch]), torch.stack([b[1] for b in batch]),
            torch.tensor([b[2] for b in batch], dtype=torch.long),
            torch.tensor([b[3] for b in batch], dtype=torch.float32), [b[4] for b in batch])


def train_epoch(model, train_loader, optimizer, criterion, device, task="classification", grad_clip=1.0):
    """one epoch of ``run_fmri_v11.py:430-449``.  With a ``FusedAdamW`` the clip is part of its step
    (``max_grad_norm``); with any other optimizer ``clip_grad_norm_`` runs here as in the reference."""
    from .optim import FusedAdamW
    model.train()
    total = 0.0
    for activation, connectivity, class_labels, reg_labels, _ in train_loader:
        labels = (class_labels if task == "classification" else reg_labels).to(device)
        optimizer.zero_grad()
        loss = criterion(model(activation.to(device), connectivity.to(device)), labels)
        loss.backward()
        if isinstance(optimizer, FusedAdamW):
            optimizer.max_grad_norm = float(grad_clip)
        elif grad_clip > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        optimizer.step()
        total += loss.item()
    return total / len(train_loader)


def classification_metrics(targets, preds, probs=None, num_classes=2):
    """Accuracy / weighted F1, precision, recall (+ AUC for two classes; 0.5 when undefined) - the
    metric dict of ``run_fmri_v11.py:482-493`` and ``_test_bridge.py:810-819``."""
    from sklearn.metrics import accuracy_score, f1_score, precision_score, recall_score, roc_auc_score
    m = {"Accuracy": accuracy_score(targets, preds),
         "F1": f1_score(targets, preds, average="weighted", zero_division=0),
         "Precision": precision_score(targets, preds, average="weighted", zero_division=0),
         "Recall": recall_score(targets, preds, average="weighted", zero_division=0)}
    if probs is not None and num_classes == 2:
        try:
            m["AUC"] = roc_auc_score(targets, probs[:, 1])
        except Exception:
            m["AUC"] = 0.5
    return m


@torch.no_grad()
def evaluate(model, data_loader, device, task="classification", num_classes=2):
    """(``run_fmri_v11.py:452-504``) -> (metrics, targets, probs) or, for regression, (metrics, targets, preds)."""
    import numpy as np
    model.eval()
    preds, targets, probs = [], [], []
    for activation, connectivity, class_labels, reg_labels, _ in data_loader:
        out = model(activation.to(device), connectivity.to(device)).float()
        if task == "classification":
            probs.append(torch.softmax(out, dim=1).cpu().numpy())
            preds.append(out.argmax(dim=1).cpu().numpy())
            targets.append(class_labels.numpy())
        else:
            preds.append(out.reshape(-1).cpu().numpy())
            targets.append(reg_labels.numpy())
    preds, targets = np.concatenate(preds), np.concatenate(targets)
    if task == "classification":
        probs = np.concatenate(probs)
        return classification_metrics(targets, preds, probs, num_classes), targets, probs
    from sklearn.metrics import mean_absolute_error, mean_squared_error, r2_score
    mse = mean_squared_error(targets, preds)
    return ({"MSE": mse, "RMSE": float(np.sqrt(mse)), "MAE": mean_absolute_error(targets, preds),
             "R2": r2_score(targets, preds)}, targets, preds)


def _read_numeric_csv(path):
    import numpy as np
    import pandas as pd
    df = pd.read_csv(path)
    if "Subject" in df.columns:
        df = df.drop("Subject", axis=1)
    return np.nan_to_num(df.values.astype(np.float32), nan=0.0)


def load_activation_features(data_dir, subject_list, activation_types, agg_method="both"):
    """{subject: tensor}: per activation type, column-wise mean / std / [mean, std] over the rows of
    ``sub-N/subject_N_activation_<type>.csv``, concatenated over types (``fmri_utils.py:115-160``)."""
    import numpy as np
    from pathlib import Path
    features = {}
    for subj in subject_list:
        parts = []
        for act_type in activation_types:
            path = Path(data_dir) / f"sub-{subj}" / f"subject_{subj}_activation_{act_type}.csv"
            if not path.exists():
                continue
            try:
                data = _read_numeric_csv(path)
                if agg_method not in ("mean", "std", "both"):
                    raise ValueError(f"Unknown agg method: {agg_method}")
                stats = {"mean": [data.mean(axis=0)], "std": [data.std(axis=0)],
                         "both": [data.mean(axis=0), data.std(axis=0)]}[agg_method]
                parts.append(np.concatenate(stats))
            except Exception as e:  # unreadable file or unknown method: warning + skip, as the reference does
                logger.warning("Error loading %s: %s", path, e)
        if parts:
            features[subj] = torch.tensor(np.concatenate(parts), dtype=torch.float32)
    logger.info("fMRI activation features: %d/%d subjects", len(features), len(subject_list))
    return features


def load_connectivity_features(data_dir, subject_list, connectivity_types):
    """{subject: tensor}: flattened ``sub-N/subject_N_fdr_PPI_Connectivity_<type>.csv`` matrices,
    concatenated over types (``fmri_utils.py:163-201``)."""
    import numpy as np
    from pathlib import Path
    features = {}
    for subj in subject_list:
        parts = []
        for conn_type in connectivity_types:
            path = Path(data_dir) / f"sub-{subj}" / f"subject_{subj}_fdr_PPI_Connectivity_{conn_type}.csv"
            if not path.exists():
                continue
            try:
                parts.append(_read_numeric_csv(path).flatten())
            except Exception as e:
                logger.warning("Error loading %s: %s", path, e)
        if parts:
            features[subj] = torch.tensor(np.concatenate(parts), dtype=torch.float32)
    logger.info("fMRI connectivity features: %d/%d subjects", len(features), len(subject_list))
    return features


_SUBJECT_COLS = ("Subject", "subject", "SubjectID", "ID", "id")
_LABEL_COLS = ("Label", "label", "Outcome", "outcome", "Class", "class", "Group", "group")


def load_fmri_labels(label_path, subject_list):
    """{subject: 0/1} from the first of labels.csv / outcomes.csv / subjects_labels.csv /
    ../labels.csv that exists; string labels good/positive/yes/1 -> 1 (``fmri_utils.py:204-244``).
    With no label file the reference falls back to random dummy labels; so does this."""
    import numpy as np
    import pandas as pd
    from pathlib import Path
    label_path = Path(label_path)
    candidates = [label_path / "labels.csv", label_path / "outcomes.csv",
                  label_path / "subjects_labels.csv", label_path.parent / "labels.csv"]
    label_file = next((p for p in candidates if p.exists()), None)
    if label_file is None:
        logger.warning("No fMRI label file found. Using dummy labels.")
        return {s: int(np.random.randint(0, 2)) for s in subject_list}
    df = pd.read_csv(label_file)
    subj_col = next((c for c in _SUBJECT_COLS if c in df.columns), None)
    label_col = next((c for c in _LABEL_COLS if c in df.columns), None)
    if not subj_col or not label_col:
        raise ValueError(f"Cannot identify columns in {label_file}: {df.columns.tolist()}")
    wanted = set(subject_list)
    labels = {}
    for subj, label in zip(df[subj_col], df[label_col]):
        if int(subj) not in wanted:
            continue
        if isinstance(label, str):
            label = 1 if label.lower() in ("good", "positive", "yes", "1") else 0
        labels[int(subj)] = int(label)
    return labels


# ---------------------------------------------------------------------------
# the fMRI driver (run_fmri_v11.py:43-77, 715-934): configuration bag and the cross-validated experiment
# ---------------------------------------------------------------------------
SEED = 42            # run_fmri_v11.py:28


class fMRIConfig:
    """attribute bag of ``run_fmri_v11.py:43-77``: same names and defaults.  ``base_path`` defaults to the
    environment variable FMRI_DATA_PATH or the working directory (the reference hard-codes a Windows path); the
    three output directories are created on construction, as there."""

    def __init__(self, base_path=None):
        import os
        from pathlib import Path
        self.base_path = Path(base_path if base_path is not None else os.environ.get("FMRI_DATA_PATH", "."))
        self.data_dir = self.base_path
        self.label_path = self.base_path / "DATA" / "labels"
        self.subject_list = list(range(1, 33))
        self.activation_types = ["sensory", "AN", "LN", "cognitive", "DMN"]
        self.connectivity_types = ["DMN"]
        self.agg_method = "both"
        self.hidden_dim = 64
        self.fusion_dim = 128
        self.dropout = 0.4
        self.num_classes = 2
        self.batch_size = 8
        self.num_epochs = 100
        self.learning_rate = 1e-4
        self.weight_decay = 1e-4
        self.patience = 15
        self.n_splits = 5
        self.val_ratio = 0.15
        self.grad_clip = 1.0
        self.output_dir = Path("./results_fmri")
        self.checkpoint_dir = Path("./checkpoints_fmri")
        self.log_dir = Path("./logs_fmri")
        for d in (self.output_dir, self.checkpoint_dir, self.log_dir):
            d.mkdir(parents=True, exist_ok=True)

    def __repr__(self):
        return (f"fMRIConfig(subjects={len(self.subject_list)}, activation={self.activation_types}, "
                f"connectivity={self.connectivity_types}, agg={self.agg_method}, val_ratio={self.val_ratio})")


def run_experiment(dataset, config, task="classification", device=None, optimizer_factory=None, verbose=True):
    """The protocol of ``run_fmri_v11.py:715-934``: (stratified) K-fold over the subjects; inside every fold the
    training part is split again into train / validation (``val_ratio``, seeded per fold); the three models
    (fusion, activation only, connectivity only) are trained with AdamW + ReduceLROnPlateau(0.5, 5) on the
    VALIDATION metric (1 - F1, or -R2), early-stopped on it (``patience``), the best state is restored and scored
    once on the held-out test part.  Returns ``(results, fusion_weights_all)`` with the reference's structure:
    ``results[name]`` = list of per-fold metric dicts.  The model calls run on the HIP kernels; the optimizer is
    ``optim.FusedAdamW`` (clip + AdamW in one launch) unless ``optimizer_factory(params, lr, weight_decay)`` says
    otherwise."""
    import copy
    import numpy as np
    from sklearn.model_selection import KFold, StratifiedKFold, train_test_split
    from sklearn.utils.class_weight import compute_class_weight
    from torch.utils.data import DataLoader, Subset
    from .optim import FusedAdamW
    from .bridge_utils import WeightedCrossEntropy
    from .crossmodal_eeg_scr import _PlateauLR
    say = print if verbose else (lambda *a, **k: None)
    device = device or torch.device("cuda")
    if task == "classification":
        labels = np.array([s["class_label"] for s in dataset.samples])
        num_classes = len(np.unique(labels))
    else:
        labels = np.array([s["reg_label"] for s in dataset.samples])
        num_classes = 1
    sample = dataset[0]
    activation_dim, connectivity_dim = sample[0].shape[0], sample[1].shape[0]
    say(f"EXPERIMENT: {task.upper()}  device {device}  validation ratio {config.val_ratio}")
    say(f"  activation dim {activation_dim}, connectivity dim {connectivity_dim}, samples {len(dataset)}")
    if task == "classification":
        splits = list(StratifiedKFold(n_splits=config.n_splits, shuffle=True, random_state=SEED).split(np.zeros(len(dataset)), labels))
    else:
        splits = list(KFold(n_splits=config.n_splits, shuffle=True, random_state=SEED).split(np.zeros(len(dataset))))
    results = {"fusion": [], "activation_only": [], "connectivity_only": []}
    fusion_weights_all = []
    for fold_idx, (train_val_idx, test_idx) in enumerate(splits, 1):
        strat = labels[train_val_idx] if task == "classification" else None
        train_idx, val_idx = train_test_split(np.arange(len(train_val_idx)), test_size=config.val_ratio, stratify=strat,
                                              random_state=SEED + fold_idx)
        actual_train_idx, actual_val_idx = train_val_idx[train_idx], train_val_idx[val_idx]
        say(f"FOLD {fold_idx}/{config.n_splits}: train {len(actual_train_idx)}, val {len(actual_val_idx)}, test {len(test_idx)}")
        mk = lambda idx, shuffle: DataLoader(Subset(dataset, idx), batch_size=config.batch_size, shuffle=shuffle,   # noqa: E731
                                             collate_fn=collate_fmri)
        train_loader, val_loader, test_loader = mk(actual_train_idx, True), mk(actual_val_idx, False), mk(test_idx, False)
        if task == "classification":
            train_labels = labels[actual_train_idx]
            cw = compute_class_weight("balanced", classes=np.unique(train_labels), y=train_labels)
            criterion = WeightedCrossEntropy(torch.tensor(cw, dtype=torch.float32)).to(device)
        else:
            criterion = nn.MSELoss()
        for model_name in ("fusion", "activation_only", "connectivity_only"):
            if model_name == "fusion":
                model = fMRIFusionNet(activation_dim=activation_dim, connectivity_dim=connectivity_dim, hidden_dim=config.hidden_dim,
                                      num_classes=num_classes, dropout=config.dropout, task=task)
            elif model_name == "activation_only":
                model = fMRIActivationOnly(in_dim=activation_dim, hidden_dim=config.hidden_dim, num_classes=num_classes,
                                           dropout=config.dropout, task=task)
            else:
                model = fMRIConnectivityOnly(in_dim=connectivity_dim, hidden_dim=config.hidden_dim, num_classes=num_classes,
                                             dropout=config.dropout, task=task)
            model = model.to(device)
            if optimizer_factory is not None:
                optimizer = optimizer_factory(model.parameters(), config.learning_rate, config.weight_decay)
            else:
                optimizer = FusedAdamW(model.parameters(), lr=config.learning_rate, weight_decay=config.weight_decay)
            # ReduceLROnPlateau(mode='min', factor=0.5, patience=5) on either optimizer kind (torch's class insists on a
            # torch.optim.Optimizer; _PlateauLR is the same rule on ``param_groups``)
            scheduler = _PlateauLR(optimizer, factor=0.5, patience=5)
            best_val_metric, best_state, patience_counter = -np.inf, None, 0
            for epoch in range(1, config.num_epochs + 1):
                train_loss = train_epoch(model, train_loader, optimizer, criterion, device, task, config.grad_clip)
                val_metrics, _, _ = evaluate(model, val_loader, device, task, num_classes)
                current = val_metrics["F1"] if task == "classification" else val_metrics["R2"]
                scheduler.step(1 - current if task == "classification" else -current)
                if epoch % 10 == 0:
                    say(f"  {model_name} epoch {epoch:3d}: loss {train_loss:.4f}, val {current:.4f}")
                if current > best_val_metric:
                    best_val_metric, best_state, patience_counter = current, copy.deepcopy(model.state_dict()), 0
                else:
                    patience_counter += 1
                if patience_counter >= config.patience:
                    say(f"  {model_name}: early stopping at epoch {epoch}")
                    break
            if best_state:
                model.load_state_dict(best_state)
            test_metrics, _, _ = evaluate(model, test_loader, device, task, num_classes)
            results[model_name].append(test_metrics)
            if model_name == "fusion":
                fusion_weights_all.append(model.get_fusion_weights())
            say(f"  {model_name} test: " + ", ".join(f"{k}={v:.4f}" for k, v in test_metrics.items()))
    return results, fusion_weights_all

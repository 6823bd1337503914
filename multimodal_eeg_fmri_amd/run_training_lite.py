"""V4-Lite tri-modal training entry point on the MI355X HIP path.

Counterpart of the reference's ``EEG_CODE/run_training_lite.py`` with the same
call surface (``main``, ``ImprovedTriModalFusionNetLite``, ``collate_balanced``,
``aggregate_features``, ``load_labels``, ``normalize_modality``,
``vec_upper_triangle``) and the same protocol: StratifiedGroupKFold (:431-433),
LabelSmoothingCrossEntropy(0.1) + AdamW(wd 0.01) + CosineAnnealingWarmup(3) +
EarlyStopping(15) (:465-468), grad-clip 1.0 (:487), best-F1 state restore
(:510-520).  The reference's ``main()`` cannot run against its own ``Config``
 and reads private clinical ``.mat`` files; here ``main()``
drives the same loop on the synthetic stand-in described by
``Config.synthetic`` (BASELINE config #1: 8 ch x 256 samples, conn 459).
"""
from __future__ import annotations

from collections import defaultdict
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


def main(config: Optional[Config] = None, max_epochs: Optional[int] = None):
    """5-fold cross-validated training of the Lite tri-modal net (synthetic data)."""
    from sklearn.metrics import accuracy_score, f1_score
    from sklearn.model_selection import StratifiedGroupKFold
    from torch.utils.data import DataLoader, Subset

    if not torch.cuda.is_available():
        raise RuntimeError("run_training_lite.main(): the HIP path needs an MI355X (no CPU fallback)")
    device = torch.device("cuda")
    set_seed(42)
    config = config or Config(None)
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

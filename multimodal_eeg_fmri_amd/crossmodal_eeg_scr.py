"""Training-side classes of the reference's EEG notebook (``EEG_CODE/CrossModal_EEG_scr.ipynb``) on the
MI355X HIP path: ``PerFoldNormalizer`` (cell 19), ``FocalLoss`` (cell 20), ``ImprovedTriModalFusionNet`` /
``ImprovedSmartFusionNet`` (cells 21-22), ``FlexibleTrainer`` (cell 23) and ``collate_trimodal`` (cell 24).

SURVEY.md §8 (f).1 / (f).3: same names, constructor arguments, batch conventions and checkpoint
container (``epoch / model_state_dict / optimizer_state_dict / scheduler_state_dict / metrics``) as
the notebook, so ``best_trimodal_fold*.pt`` files move between the two.  Model forward/backward, the
losses and the optimizer step run in the HIP kernels; there is no CPU path.
"""
from __future__ import annotations

import logging
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .bridge_utils import ImprovedTriModalFusionNet, WeightedCrossEntropy
from .crossmodal_v4_enhancements import EnhancedSmartFusionNetV4, get_fusion_weights_from_model
from .optim import FusedAdamW


class PerFoldNormalizer:
    """global mean / std over every value of the training subjects' entries (keys are tuples whose
    first element is the subject), applied to all entries (cell 19)."""

    def __init__(self):
        self.stats = {}

    def fit_on_indices(self, data_dict, train_indices, subject_array):
        train_subjects = set(np.asarray(subject_array)[train_indices])
        vals = [np.asarray(v).flatten() for k, v in data_dict.items() if k[0] in train_subjects]
        flat = np.concatenate(vals)
        self.stats["mean"] = np.mean(flat)
        self.stats["std"] = np.std(flat) + 1e-8

    def transform(self, data_dict):
        return {k: (v - self.stats["mean"]) / self.stats["std"] for k, v in data_dict.items()}


class FocalLoss(nn.Module):
    """alpha (1 - p_t)^gamma CE (cell 20), one kernel forward + gradient."""

    def __init__(self, alpha: float = 0.25, gamma: float = 2.0, reduction: str = "mean"):
        super().__init__()
        self.alpha, self.gamma, self.reduction = alpha, gamma, reduction

    def forward(self, inputs, targets):
        red = self.reduction if self.reduction in ("mean", "sum") else "none"
        return ops.focal_loss(inputs, targets, self.alpha, self.gamma, red)


class ImprovedSmartFusionNet(nn.Module):
    """two-modality checkpoint wrapper (cell 22): keys carry the ``model.`` prefix;
    ``return_feats`` -> dict(logits, gates, fused_feats)."""

    def __init__(self, in_pw_dim, in_erp_dim, fusion_dim=128, num_classes=2, dropout=0.4,
                 num_transformer_layers=2, num_heads=4, use_cross_attention=True):
        super().__init__()
        self.model = EnhancedSmartFusionNetV4(erp_channels=in_erp_dim, pw_channels=in_pw_dim, hidden_dim=fusion_dim,
                                              num_classes=num_classes, dropout=dropout,
                                              num_transformer_layers=num_transformer_layers, num_heads=num_heads,
                                              use_cross_attention=use_cross_attention)
        self.use_cross_attention = use_cross_attention
        self.fusion_weight_history = []

    def forward(self, erp, pw, return_feats=False):
        if return_feats:
            logits, gates, fused = self.model(erp, pw, return_fusion_weights=True, return_fused_feats=True)
            return {"logits": logits, "gates": gates, "fused_feats": fused}
        return self.model(erp, pw)

    def get_fusion_weights(self):
        return get_fusion_weights_from_model(self.model)

    def track_fusion_weights(self):
        w = self.get_fusion_weights()
        if w:
            self.fusion_weight_history.append(w)

    def get_weight_history(self):
        return self.fusion_weight_history


def _channels_first(x):
    """(B, T, C) with T > C -> (B, C, T): the notebook's orientation fix for ERP / power inputs."""
    if x is not None and x.dim() == 3 and x.shape[1] > x.shape[2]:
        return x.transpose(1, 2)
    return x


def collate_trimodal(batch):
    """(cell 24) 5-tuples (erp, pw, conn, subject, y) or 4-tuples (erp, pw, subject, y)."""
    n = len(batch[0])
    if n not in (4, 5):
        raise ValueError(f"Unexpected batch element length: {n}")
    cols = list(zip(*batch))
    erps, pws = _channels_first(torch.stack(cols[0])), _channels_first(torch.stack(cols[1]))
    conns = torch.stack(cols[2]) if n == 5 else None
    return (erps, pws, conns, torch.tensor(cols[-2], dtype=torch.long), torch.tensor(cols[-1], dtype=torch.long))


class _PlateauLR:
    """``ReduceLROnPlateau(mode='min', factor, patience)`` with torch's defaults (rel threshold 1e-4, no
    cooldown, min_lr 0, eps 1e-8) over a FusedAdamW, with the same ``state_dict`` field names."""

    def __init__(self, optimizer, factor=0.5, patience=5, threshold=1e-4, eps=1e-8):
        self.optimizer, self.factor, self.patience, self.threshold, self.eps = optimizer, factor, patience, threshold, eps
        self.best, self.num_bad_epochs, self.last_epoch = float("inf"), 0, 0

    def step(self, metric):
        metric = float(metric)
        self.last_epoch += 1
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = metric, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            for g in self.optimizer.param_groups:
                new = g["lr"] * self.factor
                if g["lr"] - new > self.eps:
                    g["lr"] = new
            self.num_bad_epochs = 0

    def state_dict(self):
        return {"factor": self.factor, "patience": self.patience, "threshold": self.threshold, "eps": self.eps,
                "best": self.best, "num_bad_epochs": self.num_bad_epochs, "last_epoch": self.last_epoch,
                "mode": "min", "threshold_mode": "rel", "cooldown": 0, "cooldown_counter": 0, "min_lrs": [0.0]}

    def load_state_dict(self, sd):
        for k in ("factor", "patience", "threshold", "eps", "best", "num_bad_epochs", "last_epoch"):
            if k in sd:
                setattr(self, k, sd[k])


class FlexibleTrainer:
    """(cell 23) one model + criterion + AdamW + ReduceLROnPlateau(0.5, 5); ``modality`` selects how a
    batch is unpacked and which inputs the model receives."""

    def __init__(self, model: nn.Module, device: Optional[torch.device] = None, lr: float = 1e-5,
                 weight_decay: float = 1e-5, modality: str = "fusion", class_weights: Optional[torch.Tensor] = None,
                 use_focal_loss: bool = False, logger: Optional[logging.Logger] = None):
        self.device = device or torch.device("cuda")
        self.model = model.to(self.device)
        self.modality = modality
        self.logger = logger or logging.getLogger(__name__)
        if use_focal_loss:
            self.criterion = FocalLoss(alpha=0.25, gamma=2.0)
        else:
            self.criterion = WeightedCrossEntropy(None if class_weights is None else class_weights.to(self.device))
            self.criterion.to(self.device)
        self.opt = FusedAdamW(self.model.parameters(), lr=lr, weight_decay=weight_decay)
        self.scheduler = _PlateauLR(self.opt, factor=0.5, patience=5)
        self.fusion_weights_history = []

    def _forward_model(self, erp=None, pw=None, conn=None):
        if self.modality == "trimodal":
            return self.model(erp=erp, pw=pw, conn=conn)
        if self.modality == "fusion":
            return self.model(erp=erp, pw=pw)
        if self.modality == "erponly":
            return self.model(erp=erp)
        if self.modality == "pwonly":
            return self.model(pw=pw)
        raise ValueError(f"Unknown modality: {self.modality}")

    def _unpack_batch(self, batch):
        if len(batch) == 5:
            erp, pw, conn, subj, y = batch
        elif len(batch) == 4:
            (erp, pw, subj, y), conn = batch, None
        elif len(batch) == 3:
            if self.modality == "erponly":
                (erp, subj, y), pw, conn = batch, None, None
            else:
                (pw, subj, y), erp, conn = batch, None, None
        else:
            raise ValueError(f"Batch format mismatch. Got length {len(batch)}")
        return _channels_first(erp), _channels_first(pw), conn, subj, y

    def _to_device(self, *xs):
        return [x.to(self.device) if x is not None else None for x in xs]

    def _logits(self, erp, pw, conn, return_feats):
        if isinstance(self.model, ImprovedTriModalFusionNet):
            return self.model(erp=erp, pw=pw, conn=conn, return_feats=return_feats)
        if isinstance(self.model, ImprovedSmartFusionNet):
            return self.model(erp=erp, pw=pw, return_feats=return_feats)
        return self._forward_model(erp, pw, conn)

    def train_one_epoch(self, loader, grad_clip=None):
        self.model.train()
        self.opt.max_grad_norm = float(grad_clip) if grad_clip else 0.0
        total, n = 0.0, 0
        for batch in loader:
            erp, pw, conn, _, y = self._unpack_batch(batch)
            erp, pw, conn = self._to_device(erp, pw, conn)
            self.opt.zero_grad()
            loss = self.criterion(self._logits(erp, pw, conn, False), y.to(self.device).long())
            loss.backward()
            self.opt.step()
            total += loss.item()
            n += 1
        return total / max(n, 1)

    @torch.no_grad()
    def evaluate(self, loader, n_classes: int):
        from .fmri_utils import classification_metrics
        self.model.eval()
        preds, targets, probs, gates, feats, subjects = [], [], [], [], [], []
        for batch in loader:
            erp, pw, conn, subj, y = self._unpack_batch(batch)
            erp, pw, conn = self._to_device(erp, pw, conn)
            out = self._logits(erp, pw, conn, True)
            if isinstance(out, dict):
                logits = out["logits"]
                if out.get("gates") is not None:
                    gates.append(out["gates"].float().cpu().numpy())
                if out.get("fused_feats") is not None:
                    feats.append(out["fused_feats"].float().cpu().numpy())
            else:
                logits = out
            p = torch.softmax(logits.float(), dim=1).cpu().numpy()
            probs.extend(p)
            preds.extend(np.argmax(p, axis=1))
            targets.extend(y.cpu().numpy())
            subjects.extend(subj.cpu().numpy())
        targets, preds = np.array(targets), np.array(preds)
        probs = np.array(probs) if len(probs) else np.zeros((len(preds), n_classes))
        metrics = classification_metrics(targets, preds) if len(targets) else \
            {"Accuracy": 0.0, "F1": 0.0, "Precision": 0.0, "Recall": 0.0}
        return metrics, targets, probs, feats, gates, subjects

    def track_fusion_weights(self):
        if hasattr(self.model, "get_fusion_weights"):
            self.fusion_weights_history.append(self.model.get_fusion_weights())

    def get_fusion_weights(self):
        return self.model.get_fusion_weights() if hasattr(self.model, "get_fusion_weights") else None

    def save_checkpoint(self, path: str, epoch: int, metrics: Dict):
        torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict(),
                    "optimizer_state_dict": self.opt.state_dict(),
                    "scheduler_state_dict": self.scheduler.state_dict(), "metrics": metrics}, path)
        self.logger.info("Checkpoint saved to %s", path)

    def load_checkpoint(self, path: str):
        ck = torch.load(path, map_location=self.device, weights_only=False)
        self.model.load_state_dict(ck["model_state_dict"])
        ops.weights_changed()
        self.opt.load_state_dict(ck["optimizer_state_dict"])
        self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        self.logger.info("Checkpoint loaded from %s", path)
        return ck["epoch"], ck["metrics"]

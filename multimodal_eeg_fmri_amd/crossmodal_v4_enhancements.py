"""V4-Lite EEG path + training utilities on the MI355X HIP path.

Class surface of the reference's ``EEG_CODE/crossmodal_v4_enhancements.py``:
the duplicated encoder family (:29-271, re-exported from
``enhanced_models_v4``), the full V4 classifiers EnhancedTriModalFusionNetV4 :278-388,
BiDirectionalCrossAttention :403-466, EnhancedSmartFusionNetV4 :473-570, DropPath :639-658, LabelSmoothingCrossEntropy
:665-677, EnhancedConnEncoder :684-739, HybridFusionModule :746-810,
LiteERPEncoder/LitePowerEncoder :817-877, EnhancedTriModalFusionNetV4Lite
:880-948, CosineAnnealingWarmup :1084-1112, EarlyStopping :1115-1143,
get_lite_fusion_weights :1146-1152, BalancedTriModalDataset :955-1077.  Leaf ``torch.nn`` modules are parameter
containers (identical ``state_dict``); arithmetic is in the HIP library.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from .enhanced_models_v4 import (  # noqa: F401  (reference keeps copies here)
    EnhancedERPEncoder, EnhancedPowerEncoder, LearnedFusionModule,
    PositionalEncoding, TemporalTransformerBlock)


# ----------------------------------------------------------------- utilities
def drop_path(x: torch.Tensor, drop_prob: float = 0.0, training: bool = False) -> torch.Tensor:
    """Per-sample stochastic depth; identity in eval / p == 0."""
    if drop_prob == 0.0 or not training:
        return x
    return ops.drop_path(x, drop_prob)


class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training)


class LabelSmoothingCrossEntropy(nn.Module):
    """(1-s)*NLL + s*mean(-logp), averaged over the batch."""

    def __init__(self, smoothing: float = 0.1):
        super().__init__()
        self.smoothing = smoothing
        self.confidence = 1.0 - smoothing

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return ops.smoothed_cross_entropy(pred, target, self.smoothing)


class CosineAnnealingWarmup:
    """Linear warm-up for ``warmup_epochs`` then cosine decay to ``min_lr``."""

    def __init__(self, optimizer, warmup_epochs: int, total_epochs: int,
                 min_lr: float = 1e-6, base_lr: float = None):
        self.optimizer = optimizer
        self.warmup_epochs = warmup_epochs
        self.total_epochs = total_epochs
        self.min_lr = min_lr
        self.base_lr = base_lr or optimizer.param_groups[0]["lr"]
        self.current_epoch = 0

    def _lr_at(self, epoch: int) -> float:
        if epoch <= self.warmup_epochs:
            return self.base_lr * (epoch / self.warmup_epochs)
        frac = (epoch - self.warmup_epochs) / (self.total_epochs - self.warmup_epochs)
        return self.min_lr + 0.5 * (self.base_lr - self.min_lr) * (1 + math.cos(math.pi * frac))

    def step(self) -> float:
        self.current_epoch += 1
        lr = self._lr_at(self.current_epoch)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        return lr

    def get_lr(self) -> float:
        return self.optimizer.param_groups[0]["lr"]


class EarlyStopping:
    """Stop after ``patience`` calls without a > ``min_delta`` improvement."""

    def __init__(self, patience: int = 10, min_delta: float = 0.001, mode: str = "max"):
        self.patience, self.min_delta, self.mode = patience, min_delta, mode
        self.counter = 0
        self.best_score = None
        self.should_stop = False

    def __call__(self, score) -> bool:
        if self.best_score is None:
            self.best_score = score
            return False
        better = (score > self.best_score + self.min_delta) if self.mode == "max" \
            else (score < self.best_score - self.min_delta)
        if better:
            self.best_score, self.counter = score, 0
        else:
            self.counter += 1
            self.should_stop = self.should_stop or self.counter >= self.patience
        return self.should_stop


def get_lite_fusion_weights(model):
    if hasattr(model, "get_fusion_weights"):
        return model.get_fusion_weights()
    return getattr(model, "_fusion_weights", None)


def get_fusion_weights_from_model(model):
    """{'temperature', 'erp_weight', 'pw_weight'[, 'conn_weight']} of a model with a LearnedFusionModule
    at ``.fusion`` (reference :577-602); None otherwise."""
    fusion = getattr(model, "fusion", None)
    if fusion is None or not hasattr(fusion, "fusion_logits"):
        return None
    with torch.no_grad():
        temp = fusion.temperature
        w = torch.softmax(fusion.fusion_logits / temp, dim=0).tolist()
    result = {"temperature": temp.detach().item()}
    if len(w) in (2, 3):
        result.update(zip(("erp_weight", "pw_weight", "conn_weight"), w))
    return result


def count_parameters(model: nn.Module) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


# ------------------------------------------------- full V4 classifiers
def _mlp_bn(i, o, dropout):
    return [nn.Linear(i, o), nn.BatchNorm1d(o), nn.GELU(), nn.Dropout(dropout)]


def _v4_head(hidden_dim, num_classes, dropout):
    return nn.Sequential(*_mlp_bn(hidden_dim, hidden_dim, dropout), *_mlp_bn(hidden_dim, hidden_dim // 2, dropout),
                         nn.Linear(hidden_dim // 2, num_classes))


def _flagged(logits, weights, fused, return_fusion_weights, return_fused_feats):
    out = [logits]
    if return_fusion_weights:
        out.append(weights)
    if return_fused_feats:
        out.append(fused)
    return out[0] if len(out) == 1 else tuple(out)


class EnhancedTriModalFusionNetV4(nn.Module):
    """ERP + Power transformer encoders, MLP connectivity encoder, ERP-queries-all cross attention,
    learned 3-way fusion, BN-MLP classifier (reference :278-388; same state_dict)."""

    def __init__(self, erp_channels: int, pw_channels: int, conn_features: int,
                 hidden_dim: int = 128, num_classes: int = 2, dropout: float = 0.3,
                 num_transformer_layers: int = 2, num_heads: int = 4):
        super().__init__()
        self.erp_encoder = EnhancedERPEncoder(erp_channels, hidden_dim, num_transformer_layers, num_heads, dropout)
        self.pw_encoder = EnhancedPowerEncoder(pw_channels, hidden_dim, num_transformer_layers, num_heads, dropout)
        self.conn_encoder = nn.Sequential(*_mlp_bn(conn_features, 256, dropout), *_mlp_bn(256, hidden_dim, dropout))
        self.cross_attn = nn.MultiheadAttention(hidden_dim, num_heads=num_heads, dropout=dropout, batch_first=True)
        self.fusion = LearnedFusionModule(num_modalities=3, hidden_dim=hidden_dim, use_temperature=True)
        self.classifier = _v4_head(hidden_dim, num_classes, dropout)
        self.drop_p = dropout

    def forward(self, erp, pw, conn, return_fusion_weights: bool = False, return_fused_feats: bool = False):
        logits, weights, fused = ops.trimodal_v4_forward(self, erp, pw, conn)
        return _flagged(logits, weights, fused, return_fusion_weights, return_fused_feats)


class BiDirectionalCrossAttention(nn.Module):
    """each modality attends to [erp, pw]; sigmoid-gated residual + LayerNorm (reference :403-466)."""

    def __init__(self, hidden_dim: int, num_heads: int = 4, dropout: float = 0.3):
        super().__init__()
        self.erp_to_pw_attn = nn.MultiheadAttention(hidden_dim, num_heads=num_heads, dropout=dropout, batch_first=True)
        self.pw_to_erp_attn = nn.MultiheadAttention(hidden_dim, num_heads=num_heads, dropout=dropout, batch_first=True)
        self.norm_erp = nn.LayerNorm(hidden_dim)
        self.norm_pw = nn.LayerNorm(hidden_dim)
        self.erp_gate = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.Sigmoid())
        self.pw_gate = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.Sigmoid())
        self.dropout = nn.Dropout(dropout)

    def forward(self, erp_feat, pw_feat):
        return ops.bidirectional_cross_attention_forward(self, erp_feat, pw_feat)


class EnhancedSmartFusionNetV4(nn.Module):
    """bi-modal (ERP + Power) classifier with bi-directional cross attention (reference :473-570)."""

    def __init__(self, erp_channels: int, pw_channels: int, hidden_dim: int = 128, num_classes: int = 2,
                 dropout: float = 0.4, num_transformer_layers: int = 2, num_heads: int = 4,
                 use_cross_attention: bool = True):
        super().__init__()
        self.use_cross_attention = use_cross_attention
        self.erp_encoder = EnhancedERPEncoder(erp_channels, hidden_dim, num_transformer_layers, num_heads, dropout)
        self.pw_encoder = EnhancedPowerEncoder(pw_channels, hidden_dim, num_transformer_layers, num_heads, dropout)
        if use_cross_attention:
            self.cross_attention = BiDirectionalCrossAttention(hidden_dim, num_heads=num_heads, dropout=dropout)
        self.fusion = LearnedFusionModule(num_modalities=2, hidden_dim=hidden_dim, use_temperature=True)
        self.classifier = _v4_head(hidden_dim, num_classes, dropout)
        self.drop_p = dropout

    def forward(self, erp, pw, return_fusion_weights: bool = False, return_fused_feats: bool = False):
        logits, weights, fused = ops.smart_fusion_v4_forward(self, erp, pw)
        return _flagged(logits, weights, fused, return_fusion_weights, return_fused_feats)


# ------------------------------------------------------------- data layer
class BalancedTriModalDataset(torch.utils.data.Dataset):
    """one (erp, pw, conn, label, subject) sample per subject present in all three modalities and in
    ``label_dict`` (reference :955-1077).  Feature dicts are keyed by subject or by a tuple whose
    first element is the subject; values are tensors / arrays or (feature, metadata) pairs.  Every
    entry is flattened and the subject's entries are reduced with ``agg_method`` ('mean', 'max',
    anything else = first entry)."""

    def __init__(self, erp_features: dict, pw_features: dict, conn_features: dict, label_dict: dict,
                 transform=None, agg_method: str = "mean"):
        self.transform, self.agg_method = transform, agg_method
        per_modality = [self._aggregate_by_subject(d, agg_method) for d in (erp_features, pw_features, conn_features)]
        common = set(per_modality[0]) & set(per_modality[1]) & set(per_modality[2])
        print(f"BalancedTriModalDataset: Found {len(common)} common subjects")
        print("  ERP subjects: %d, PW subjects: %d, CONN subjects: %d" % tuple(len(m) for m in per_modality))
        self.samples = [{"erp": per_modality[0][s], "pw": per_modality[1][s], "conn": per_modality[2][s],
                         "label": label_dict[s], "subject": s}
                        for s in sorted(common) if s in label_dict]
        print(f"BalancedTriModalDataset: Created {len(self.samples)} balanced samples")

    @staticmethod
    def _subject_of(key):
        return key[0] if isinstance(key, tuple) else key

    def _extract_subjects(self, features_dict):
        return {self._subject_of(k) for k in features_dict}

    def _aggregate_by_subject(self, features_dict, method="mean"):
        import numpy as np
        rows = {}
        for key, value in features_dict.items():
            feat = value[0] if isinstance(value, tuple) else value
            feat = feat.numpy() if isinstance(feat, torch.Tensor) else np.asarray(feat)
            rows.setdefault(self._subject_of(key), []).append(feat.reshape(-1))
        out = {}
        for subj, feats in rows.items():
            stacked = np.stack(feats, axis=0)
            agg = stacked.mean(axis=0) if method == "mean" else stacked.max(axis=0) if method == "max" else stacked[0]
            out[subj] = torch.tensor(agg, dtype=torch.float32)
        return out

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        s = self.samples[idx]
        erp, pw = s["erp"], s["pw"]
        if self.transform:
            erp, pw = self.transform(erp), self.transform(pw)
        return erp, pw, s["conn"], s["label"], s["subject"]


# ----------------------------------------------------------------- encoders
class MultiScaleSTFTPowerEncoder(nn.Module):
    """BASELINE config #5 front-end (north-star extension a-X3; the reference loads
    MATLAB spectra instead, eeg_data_utils.py:86-119): raw EEG (B, C, T) -> Hann STFT
    power at several window sizes -> (B, C * sum(F), frames) -> [per-sample z-score] -> EnhancedPowerEncoder.
    Semantics of the spectra = torch.stft(center=True, reflect) ** 2 per channel.  ``normalize`` (default):
    the reference feeds its power features through normalize_modality - (x - mean) / (std + 1e-8) over the
    whole feature array of a sample - before any model sees them (run_training_lite.py:48-51, 162); the
    front-end does the same in fp32 before the bf16 cast, so the encoder's operands are O(1) like every
    other modality (raw |.|^2 values span ~6 decades and cost the bf16 path a digit of parity)."""

    def __init__(self, in_channels: int, n_ffts=(64, 128), hop: int = 32, hidden_dim: int = 128,
                 num_transformer_layers: int = 2, num_heads: int = 4, dropout: float = 0.3,
                 normalize: bool = True):
        super().__init__()
        self.n_ffts, self.hop = tuple(int(n) for n in n_ffts), int(hop)
        self.normalize = bool(normalize)
        self.spec_channels = in_channels * sum(n // 2 + 1 for n in self.n_ffts)
        self.encoder = EnhancedPowerEncoder(self.spec_channels, hidden_dim, num_transformer_layers,
                                            num_heads, dropout)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.stft_power_encoder_forward(self, x)


class EnhancedConnEncoder(nn.Module):
    """conn -> 256 -> 128 (Linear-BN-GELU) -> sigmoid feature gate -> hidden."""

    def __init__(self, conn_features: int, hidden_dim: int = 96, dropout: float = 0.4):
        super().__init__()

        def block(i, o):
            return nn.Sequential(nn.Linear(i, o), nn.BatchNorm1d(o), nn.GELU(), nn.Dropout(dropout))
        self.proj1 = block(conn_features, 256)
        self.proj2 = block(256, 128)
        self.attention = nn.Sequential(nn.Linear(128, 64), nn.Tanh(),
                                       nn.Linear(64, 128), nn.Sigmoid())
        self.output = block(128, hidden_dim)
        self.drop_p = dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() > 2:
            x = x.reshape(x.size(0), -1)
        return ops.conn_encoder_forward(self, x)


class HybridFusionModule(nn.Module):
    """softmax-gated ERP/PW mix, boosted CONN, Linear(2H->H)-BN-GELU."""

    def __init__(self, hidden_dim: int, dropout: float = 0.3, conn_boost: float = 1.2):
        super().__init__()
        self.conn_boost = conn_boost
        self.erp_pw_gate = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.GELU(),
                                         nn.Dropout(dropout), nn.Linear(hidden_dim, 2),
                                         nn.Softmax(dim=-1))
        self.late_fusion = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim),
                                         nn.BatchNorm1d(hidden_dim), nn.GELU(), nn.Dropout(dropout))
        self.final_gate = nn.Parameter(torch.tensor([0.6, 0.4]))
        self.drop_p = dropout

    def forward(self, erp_feat, pw_feat, conn_feat, return_weights: bool = False):
        fused, gate = ops.hybrid_fusion_forward(self, erp_feat, pw_feat, conn_feat)
        if not return_weights:
            return fused
        final = torch.softmax(self.final_gate.detach(), dim=0)
        g = gate.detach().mean(dim=0)
        weights = {"erp_weight": g[0].item() * final[0].item(),
                   "pw_weight": g[1].item() * final[0].item(),
                   "conn_weight": final[1].item() * self.conn_boost}
        return fused, weights


class _LiteEncoder(nn.Module):
    def __init__(self, in_channels, mid, k1, k2, hidden_dim, dropout):
        super().__init__()
        self.conv_layers = nn.Sequential(
            nn.Conv1d(in_channels, mid, kernel_size=k1, padding=k1 // 2), nn.BatchNorm1d(mid),
            nn.GELU(), nn.Dropout(dropout), nn.MaxPool1d(2),
            nn.Conv1d(mid, hidden_dim, kernel_size=k2, padding=k2 // 2), nn.BatchNorm1d(hidden_dim),
            nn.GELU(), nn.Dropout(dropout), nn.AdaptiveAvgPool1d(1))
        self.output = nn.Sequential(nn.Flatten(), nn.Linear(hidden_dim, hidden_dim),
                                    nn.GELU(), nn.Dropout(dropout))
        self.drop_p = dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.lite_encoder_forward(self, x)


class LiteERPEncoder(_LiteEncoder):
    def __init__(self, in_channels: int, hidden_dim: int = 96, dropout: float = 0.4):
        super().__init__(in_channels, 48, 7, 5, hidden_dim, dropout)


class LitePowerEncoder(_LiteEncoder):
    def __init__(self, in_channels: int, hidden_dim: int = 96, dropout: float = 0.4):
        super().__init__(in_channels, 64, 5, 3, hidden_dim, dropout)


class EnhancedTriModalFusionNetV4Lite(nn.Module):
    """Lite tri-modal classifier used by ``run_training_lite``."""

    def __init__(self, erp_channels: int, pw_channels: int, conn_features: int,
                 hidden_dim: int = 96, num_classes: int = 2, dropout: float = 0.4,
                 conn_boost: float = 1.3):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.erp_encoder = LiteERPEncoder(erp_channels, hidden_dim, dropout)
        self.pw_encoder = LitePowerEncoder(pw_channels, hidden_dim, dropout)
        self.conn_encoder = EnhancedConnEncoder(conn_features, hidden_dim, dropout)
        self.fusion = HybridFusionModule(hidden_dim, dropout, conn_boost)
        self.classifier = nn.Sequential(
            nn.Linear(hidden_dim, hidden_dim // 2), nn.BatchNorm1d(hidden_dim // 2),
            nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden_dim // 2, num_classes))
        self.drop_p = dropout
        self._fusion_weights = None

    def forward(self, erp, pw, conn, return_fusion_weights: bool = False,
                return_fused_feats: bool = False):
        e = self.erp_encoder(erp)
        p = self.pw_encoder(pw)
        c = self.conn_encoder(conn)
        weights = None
        if return_fusion_weights:
            fused, weights = self.fusion(e, p, c, return_weights=True)
            self._fusion_weights = weights
        else:
            fused = self.fusion(e, p, c)
        logits = ops.bn_classifier_forward(self.classifier, fused, self.drop_p, self.training)
        out = [logits]
        if return_fusion_weights:
            out.append(weights)
        if return_fused_feats:
            out.append(fused)
        return out[0] if len(out) == 1 else tuple(out)

    def get_fusion_weights(self):
        return self._fusion_weights

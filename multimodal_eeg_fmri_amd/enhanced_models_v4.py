"""EEG temporal-encoder family on the MI355X HIP path.

Drop-in class surface for the reference's ``EEG_CODE/enhanced_models_v4.py``
(PositionalEncoding :30-55, TemporalTransformerBlock :58-107,
EnhancedERPEncoder :114-193, EnhancedPowerEncoder :196-285,
LearnedFusionModule :420-488): same class names, constructor signatures,
``forward`` conventions and ``state_dict`` keys/shapes, so the reference's
``best_*_fold*.pt`` checkpoints load unchanged.

The ``torch.nn`` leaf modules created in the constructors are *parameter
containers only* (they fix key names, shapes and the default initialisation);
``forward`` never calls them.  All arithmetic runs in the hand-written gfx950
kernels behind ``libmmeeg_hip.so`` (see ``ops.py`` / ``include/mmeeg_hip.h``);
there is no CPU or eager-PyTorch fallback: a CPU tensor or a missing library
raises.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops


def _drop(p: float) -> nn.Dropout:
    return nn.Dropout(p)


class PositionalEncoding(nn.Module):
    """Sinusoidal table ``pe`` (max_len, 1, d_model) added to the sequence.

    Reference quirk kept (enhanced_models_v4.py:49): a 3-D input is treated as
    batch-first only when ``x.size(1) != 1``.
    """

    def __init__(self, d_model: int, max_len: int = 5000, dropout: float = 0.1):
        super().__init__()
        self.dropout = _drop(dropout)
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32)
                         * (-math.log(10000.0) / d_model))
        table = torch.zeros(max_len, 1, d_model)
        table[:, 0, 0::2] = torch.sin(pos * freq)
        table[:, 0, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.add_positional(x, self.pe, self.dropout.p, self.training)


class TemporalTransformerBlock(nn.Module):
    """Pre-norm MHA + FFN block; ``x`` is the fp32 residual stream (B, L, d)."""

    def __init__(self, d_model: int, nhead: int = 4, dim_feedforward: int = 512,
                 dropout: float = 0.1, activation: str = "gelu"):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout,
                                               batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = _drop(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = _drop(dropout)
        self.dropout2 = _drop(dropout)
        self.activation = nn.GELU() if activation == "gelu" else nn.ReLU()
        self._act = "gelu" if activation == "gelu" else "relu"
        self.nhead = nhead

    def forward(self, x: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        # ``mask`` = nn.MultiheadAttention's attn_mask (reference :98): (L, L) or (batch * heads, L, L), boolean
        # (True = not allowed) or additive float; the encoders themselves never pass one (reference :169-193)
        return ops.transformer_block(x, self, self.training, mask)


def _transformer_stack(hidden_dim, layers, heads, dropout):
    return nn.ModuleList([
        TemporalTransformerBlock(hidden_dim, nhead=heads,
                                 dim_feedforward=hidden_dim * 4, dropout=dropout)
        for _ in range(layers)])


def _pool_proj(hidden_dim, dropout):
    return nn.Sequential(nn.AdaptiveAvgPool1d(1), nn.Flatten(),
                         nn.Linear(hidden_dim, hidden_dim), nn.GELU(), _drop(dropout))


class EnhancedERPEncoder(nn.Module):
    """(B, C, T) fp32 -> (B, hidden_dim): 3x [Conv1d-BN-GELU] (+MaxPool2 after
    the second) -> PE -> N transformer blocks -> mean over time -> Linear-GELU."""

    def __init__(self, in_channels: int, hidden_dim: int = 128,
                 num_transformer_layers: int = 2, num_heads: int = 4,
                 dropout: float = 0.3):
        super().__init__()
        self.conv_layers = nn.Sequential(
            nn.Conv1d(in_channels, 64, kernel_size=7, padding=3), nn.BatchNorm1d(64),
            nn.GELU(), _drop(dropout),
            nn.Conv1d(64, 128, kernel_size=5, padding=2), nn.BatchNorm1d(128),
            nn.GELU(), nn.MaxPool1d(2), _drop(dropout),
            nn.Conv1d(128, hidden_dim, kernel_size=3, padding=1), nn.BatchNorm1d(hidden_dim),
            nn.GELU(), _drop(dropout))
        self.pos_encoder = PositionalEncoding(hidden_dim, dropout=dropout)
        self.transformer_layers = _transformer_stack(hidden_dim, num_transformer_layers,
                                                     num_heads, dropout)
        self.output_proj = _pool_proj(hidden_dim, dropout)
        self.drop_p = dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.erp_encoder_forward(self, x)


class EnhancedPowerEncoder(nn.Module):
    """Multi-scale k=3/5/7 Conv1d front-end -> 1x1 fusion conv -> same tail."""

    def __init__(self, in_channels: int, hidden_dim: int = 128,
                 num_transformer_layers: int = 2, num_heads: int = 4,
                 dropout: float = 0.3):
        super().__init__()

        def scale(k):
            return nn.Sequential(nn.Conv1d(in_channels, 64, kernel_size=k, padding=k // 2),
                                 nn.BatchNorm1d(64), nn.GELU())
        self.conv_scale1 = scale(3)
        self.conv_scale2 = scale(5)
        self.conv_scale3 = scale(7)
        self.fusion = nn.Sequential(nn.Conv1d(192, hidden_dim, kernel_size=1),
                                    nn.BatchNorm1d(hidden_dim), nn.GELU(), _drop(dropout))
        self.pos_encoder = PositionalEncoding(hidden_dim, dropout=dropout)
        self.transformer_layers = _transformer_stack(hidden_dim, num_transformer_layers,
                                                     num_heads, dropout)
        self.output_proj = _pool_proj(hidden_dim, dropout)
        self.drop_p = dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.power_encoder_forward(self, x)


class LearnedFusionModule(nn.Module):
    """0.5*softmax(logits/T) + 0.5*softmax(gate_net(cat)/T) weighted sum."""

    def __init__(self, num_modalities: int, hidden_dim: int,
                 use_temperature: bool = True, init_temperature: float = 1.0):
        super().__init__()
        self.num_modalities = num_modalities
        self.use_temperature = use_temperature
        self.fusion_logits = nn.Parameter(torch.ones(num_modalities))
        if use_temperature:
            self.temperature = nn.Parameter(torch.tensor(init_temperature))
        else:
            self.register_buffer("temperature", torch.tensor(1.0))
        self.gate_net = nn.Sequential(nn.Linear(hidden_dim * num_modalities, hidden_dim),
                                      nn.GELU(), _drop(0.2),
                                      nn.Linear(hidden_dim, num_modalities))

    def forward(self, modality_features: List[torch.Tensor], return_weights: bool = False):
        fused, w = ops.learned_fusion(self, list(modality_features), self.training)
        return (fused, w) if return_weights else fused

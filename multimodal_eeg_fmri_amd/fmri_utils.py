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

import torch
import torch.nn as nn

from . import ops


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

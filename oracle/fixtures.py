"""TEST INFRASTRUCTURE ONLY.  Deterministic model/input builders shared by
oracle/make_goldens.py (which applies them to the REFERENCE classes) and by
tests/ (which apply them to the product classes), so both sides hold
bit-identical weights without shipping them.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

_NORMS = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


def seeded_randn(seed: int, *shape) -> torch.Tensor:
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def perturb_norm_state(model: nn.Module, seed: int) -> None:
    """Make eval-mode BatchNorm non-trivial: fresh running stats are (0, 1) and
    affine (1, 0), which would hide a wrong BN fold.  Deterministic in module
    order, so it gives the same values on reference and product classes."""
    g = torch.Generator().manual_seed(int(seed) + 7919)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, _NORMS):
                n = m.num_features
                m.running_mean.copy_(0.2 * torch.randn(n, generator=g))
                m.running_var.copy_(0.5 + torch.rand(n, generator=g))
                m.weight.copy_(0.8 + 0.4 * torch.rand(n, generator=g))
                m.bias.copy_(0.1 * torch.randn(n, generator=g))
            elif isinstance(m, nn.LayerNorm):
                n = m.normalized_shape[0]
                m.weight.copy_(0.8 + 0.4 * torch.rand(n, generator=g))
                m.bias.copy_(0.1 * torch.randn(n, generator=g))


def build(cls, seed: int, *args, perturb: bool = True, **kwargs) -> nn.Module:
    torch.manual_seed(int(seed))
    model = cls(*args, **kwargs)
    if perturb:
        perturb_norm_state(model, seed)
    return model


def checksum(model: nn.Module) -> np.ndarray:
    """(n_tensors, 2) float64: sum and abs-sum of every floating state tensor."""
    rows = []
    for _, v in model.state_dict().items():
        if v.is_floating_point():
            d = v.double()
            rows.append((d.sum().item(), d.abs().sum().item()))
    return np.array(rows, dtype=np.float64)


def grad_summary(model: nn.Module, head: int = 64) -> dict:
    """Per-parameter gradient pins: L2 norm, sum and the first ``head`` values."""
    out = {}
    names, norms, sums = [], [], []
    for n, p in model.named_parameters():
        g = p.grad.detach().double().flatten()
        names.append(n)
        norms.append(g.norm().item())
        sums.append(g.sum().item())
        out["ghead::" + n] = g[:head].float().numpy()
    out["gnames"] = np.array(names)
    out["gnorms"] = np.array(norms)
    out["gsums"] = np.array(sums)
    return out

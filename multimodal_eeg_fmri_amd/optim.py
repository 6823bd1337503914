"""Flat parameter bucket + fused clip/AdamW (``mm_sumsq`` + ``mm_adamw_clip``).

Replaces the reference's ``clip_grad_norm_(max_norm=1.0)`` + ``torch.optim.AdamW.step()``
pair (run_training_lite.py:487-488, _test_bridge.py:784-786) with two kernel
launches over one contiguous fp32 buffer."""
from __future__ import annotations

import torch

from . import _hip, ops


class FlatBucket:
    """All trainable parameters (and their gradients / Adam moments) as single
    contiguous fp32 buffers; ``param.data`` and ``param._mm_grad`` are views, so
    kernels accumulate gradients in place and one collective covers the step."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.n = n
        self.p = torch.empty(n, dtype=torch.float32, device=dev)
        self.g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.state = torch.zeros(8 + 1024, dtype=torch.float32, device=dev)   # MM_OPT_STATE_FLOATS
        off = 0
        for p in self.params:
            k = p.numel()
            self.p[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.p[off:off + k].view(p.shape)
            p._mm_grad = self.g[off:off + k].view(p.shape)
            off += k

    def zero_grad(self):
        self.g.zero_()
        for p in self.params:
            p.grad = None

    def absorb_autograd_grads(self):
        """parameters whose gradient came back through autograd (not a kernel sink)"""
        for p in self.params:
            if p.grad is not None:
                p._mm_grad.add_(p.grad)
                p.grad = None


class FusedAdamW:
    """torch.optim.AdamW-like surface (``param_groups[0]['lr']``, ``zero_grad``, ``step``)
    with the global-norm clip folded in (``max_grad_norm`` <= 0 disables it)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
                 max_grad_norm: float = 0.0):
        self._all = list(params)                  # torch's state_dict numbers EVERY param of the group
        self.bucket = FlatBucket(self._all)
        self.param_groups = [{"lr": lr, "params": self.bucket.params}]
        self.betas, self.eps, self.weight_decay, self.max_grad_norm = betas, eps, weight_decay, max_grad_norm
        ops.weights_changed()

    def zero_grad(self, set_to_none: bool = True):
        self.bucket.zero_grad()

    def step(self):
        b = self.bucket
        b.absorb_autograd_grads()
        b.state[2] = float(self.param_groups[0]["lr"])
        _hip.call("mm_sumsq", b.g, b.state, b.n)
        _hip.call("mm_adamw_clip", b.p, b.g, b.m, b.v, b.state, b.n, self.betas[0], self.betas[1],
                  self.eps, self.weight_decay, float(self.max_grad_norm), 1.0, 0, None)
        ops.weights_changed()

    # -- checkpoint drop-in: the dict layout of torch.optim.AdamW.state_dict(), so a
    # FlexibleTrainer checkpoint (EEG notebook cell 23: 'optimizer_state_dict') moves both ways.
    def _slices(self):
        off = 0
        index = {id(p): i for i, p in enumerate(self._all)}
        for p in self.bucket.params:
            k = p.numel()
            yield index[id(p)], p, slice(off, off + k)
            off += k

    def state_dict(self):
        b = self.bucket
        step = float(b.state[0].item())
        state = {i: {"step": torch.tensor(step), "exp_avg": b.m[sl].view(p.shape).clone(),
                     "exp_avg_sq": b.v[sl].view(p.shape).clone()} for i, p, sl in self._slices()}
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.betas), "eps": self.eps,
                 "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True,
                 "params": list(range(len(self._all)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self._all):
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of "
                             "optimizer's group")
        if group.get("amsgrad"):
            raise ValueError("FusedAdamW has no amsgrad state")
        b = self.bucket
        self.param_groups[0]["lr"] = group["lr"]
        self.betas, self.eps, self.weight_decay = tuple(group["betas"]), group["eps"], group["weight_decay"]
        steps = set()
        b.m.zero_(); b.v.zero_()
        for i, p, sl in self._slices():
            st = sd["state"].get(i)
            if st is None:
                continue
            b.m[sl].copy_(st["exp_avg"].reshape(-1))
            b.v[sl].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(float(st["step"]))
        if len(steps) > 1:
            raise ValueError("FusedAdamW keeps one step counter; the loaded per-parameter steps differ")
        b.state[0] = steps.pop() if steps else 0.0

    @property
    def last_grad_norm(self) -> float:
        return float(self.bucket.state[4])

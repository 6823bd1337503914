#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Round-2 additions to tests/golden/ (same rules as make_goldens.py: the
REFERENCE is imported read-only from /root/reference in the build container, weights are rebuilt from seeds
and checksummed, only small arrays are stored):

  a1a2_standalone.npz      PositionalEncoding.forward (batch-first and the (seq, 1, d) quirk branch) and
                           TemporalTransformerBlock.forward(x, mask) for mask = None / boolean causal /
                           additive float, eval outputs + train-mode (dropout 0) gradients
  a3_erp_train_grads_c2.npz  EnhancedERPEncoder train-mode forward + every gradient at the C2 SHAPE
                           (64 ch x 1024 samples, B = 2): the shape bench.py times

Run:  python oracle/make_goldens_r2.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_functional as RF  # noqa: E402
from oracle.fixtures import build, checksum, grad_summary, seeded_randn  # noqa: E402
from oracle.make_goldens import OUT, _close, _import_reference, _np, _same_state  # noqa: E402


def masks(L):
    causal = torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1)
    band = seeded_randn(305, L, L) * 0.5                      # additive float mask with a banded -inf part
    band = band.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=17), float("-inf"))
    return {"none": None, "causal": causal, "float": band}


def main():
    cv4, _, _ = _import_reference()
    import multimodal_eeg_fmri_amd.enhanced_models_v4 as ours_e
    torch.set_num_threads(8)
    fx = {}
    # ------------------------------------------------------------------ a1
    ref = cv4.PositionalEncoding(128, dropout=0.1).eval()
    our = ours_e.PositionalEncoding(128, dropout=0.1).eval()
    assert torch.equal(ref.pe, our.pe)
    xb = seeded_randn(301, 2, 96, 128)                        # batch-first
    xs = seeded_randn(302, 40, 1, 128)                        # (seq, batch = 1, d): the size(1) == 1 branch
    with torch.no_grad():
        yb, ys = ref(xb), ref(xs)
    _close(RF.positional_encoding({"pe": ref.pe}, "", xb), yb, "a1 batch-first")
    _close(RF.positional_encoding({"pe": ref.pe}, "", xs), ys, "a1 seq-first")
    fx.update(pe_x_seeds=np.array([301, 302]), pe_out_bf=_np(yb), pe_out_sf=_np(ys))
    # ------------------------------------------------------------------ a2
    L = 96
    ref = build(cv4.TemporalTransformerBlock, 41, 128, 4, 512, 0.1).eval()
    our = build(ours_e.TemporalTransformerBlock, 41, 128, 4, 512, 0.1).eval()
    _same_state(ref, our, "TemporalTransformerBlock")
    x = seeded_randn(303, 2, L, 128)
    fx.update(blk_seed=41, blk_x_seed=303, blk_cks=checksum(ref), mask_float=_np(masks(L)["float"]))
    for tag, m in masks(L).items():
        with torch.no_grad():
            y = ref(x, m)
            yo = RF.transformer_block(our.state_dict(), "", x, 4, mask=m)
        _close(yo, y, f"a2 eval mask={tag}", 5e-6)
        fx[f"blk_out_{tag}"] = _np(y)
    gy = seeded_randn(304, 2, L, 128)
    for tag, m in masks(L).items():
        ref = build(cv4.TemporalTransformerBlock, 42, 128, 4, 512, 0.0).train()
        xg = x.clone().requires_grad_(True)
        ref(xg, m).backward(gy)
        sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in ref.state_dict().items()}
        xo = x.clone().requires_grad_(True)
        RF.transformer_block(sd, "", xo, 4, mask=m).backward(gy)
        _close(xo.grad, xg.grad, f"a2 dx mask={tag}", 2e-5)
        for n, p in ref.named_parameters():
            _close(sd[n].grad, p.grad, f"a2 d{n} mask={tag}", 5e-5)
        fx[f"blk_dx_{tag}"] = _np(xg.grad)
        fx.update({f"blk_{tag}_" + k: v for k, v in grad_summary(ref, head=16).items()})
    fx.update(blk_train_seed=42, blk_gy_seed=304)
    np.savez_compressed(os.path.join(OUT, "a1a2_standalone.npz"), **fx)

    # -------------------------------------------------- a3 train-mode gradients at the C2 shape
    B, C, T = 2, 64, 1024
    ref = build(cv4.EnhancedERPEncoder, 23, C, 128, 2, 4, 0.0).train()
    our = build(ours_e.EnhancedERPEncoder, 23, C, 128, 2, 4, 0.0).train()
    _same_state(ref, our, "a3 train c2")
    x = seeded_randn(125, B, C, T).requires_grad_(True)
    gy = seeded_randn(126, B, 128)
    cks_before = checksum(ref)
    y = ref(x)
    y.backward(gy)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in our.state_dict().items()}
    xo = x.detach().clone().requires_grad_(True)
    yo = RF.erp_encoder(sd, xo, train=True)
    yo.backward(gy)
    _close(yo, y, "a3 c2 train out", 5e-6)
    _close(xo.grad, x.grad, "a3 c2 dx", 2e-5)
    for n, p in ref.named_parameters():
        _close(sd[n].grad, p.grad, "a3 c2 d" + n, 5e-5)
    fx = {"seed": 23, "x_seed": 125, "gy_seed": 126, "shape": np.array([B, C, T]), "out": _np(y),
          "dx_t8": _np(x.grad[:, :, ::8]), "dx_norm": np.array(x.grad.double().norm().item()),
          "cks": cks_before, "cks_after": checksum(ref)}
    fx.update(grad_summary(ref, head=32))
    np.savez_compressed(os.path.join(OUT, "a3_erp_train_grads_c2.npz"), **fx)
    print("round-2 goldens written to", OUT)


if __name__ == "__main__":
    main()

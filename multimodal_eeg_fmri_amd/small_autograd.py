"""torch.autograd wrappers of the small fp32 kernels (dense layers, BatchNorm1d
over a (B, N) batch, gates, label-smoothing CE) used to TRAIN the tabular /
Lite models on the HIP path: V4-Lite tri-modal net (crossmodal_v4_enhancements.py
:684-948), fMRI MLPs (fmri_utils.py:23-108).  Every forward/backward FLOP is a
C-ABI kernel; gradients go to ``param._mm_grad`` sinks when present."""
from __future__ import annotations

import torch

from . import _hip, ops
from .autograd import GradBag, _compact, conv_bn_act_bwd
from .ops import ACT, REPL, _F32, _empty, _zeros


def _f(t):
    return t.float().contiguous()


class SmallLinearFn(torch.autograd.Function):
    """y = dropout(act(x W^T + b)), fp32 rows."""

    @staticmethod
    def forward(ctx, x, W, b, act, drop_p):
        x = _f(x)
        B, K = x.shape
        N = W.shape[0]
        seed = ops._next_seed() if drop_p > 0 else 0
        y = _empty((B, N), _F32, x)
        pre = _empty((B, N), _F32, x)
        _hip.call("mm_small_linear_fwd", x, W, b, None, None, y, pre, B, K, N, ACT[act], float(drop_p), seed, ops.EP())
        ctx.save_for_backward(x, pre)
        ctx.W, ctx.b, ctx.meta = W, b, (act, float(drop_p), seed, x.requires_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        act, p, seed, _ = ctx.meta
        B, K = x.shape
        N = W.shape[0]
        dz = _empty((B, N), _F32, dy)
        _hip.call("mm_act_bwd_f32", _f(dy), pre, dz, B * N, ACT[act], p, seed, ops.EP())
        bag = GradBag()
        dx = _empty((B, K), _F32, dy)
        _hip.call("mm_small_linear_bwd", dz, x, W, dx, bag.target(W), bag.target(b), B, K, N)
        return dx, bag.result(W), (bag.result(b) if b is not None else None), None, None


class BNRowsActFn(torch.autograd.Function):
    """y = dropout(act(BatchNorm1d_train(x))) on fp32 (B, N); updates running stats."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, act, drop_p):
        x = _f(x)
        B, N = x.shape
        stats = _zeros((REPL, 2, N), x)
        _hip.call("mm_colstats", x, stats, B, N)                 # replica 0
        out4 = ops.bn_finalize_train(bn, stats, B)
        seed = ops._next_seed() if drop_p > 0 else 0
        y = _empty((B, N), _F32, x)
        _hip.call("mm_bn_act_fwd", x, out4[0], out4[1], None, None, y, 1, B, N, ACT[act], 1, 1,
                  float(drop_p), seed, 0.0, 0, ops.EP())
        ctx.save_for_backward(x, out4)
        ctx.bn, ctx.meta = bn, (act, float(drop_p), seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, out4 = ctx.saved_tensors
        act, p, seed = ctx.meta
        bn = ctx.bn
        B, N = x.shape
        dy = _f(dy)
        sums = _zeros((REPL, 2, N), x)
        args = (1, B, N, ACT[act], 1, 1, p, seed, 0.0, 0, ops.EP())
        _hip.call("mm_bn_act_bwd_reduce", x, out4, None, dy, sums, *args)
        sc = _compact(sums, 2 * N)
        dx = _empty((B, N), _F32, x)
        _hip.call("mm_bn_act_bwd_apply", x, out4, None, dy, sc, None, dx, *args, 1)
        bag = GradBag()
        gg, gb = bag.target(bn.weight), bag.target(bn.bias)
        if gb is not None:
            _hip.call("mm_reduce_replicas", sc, gb, N, 1, N)
        if gg is not None:
            _hip.call("mm_reduce_replicas", sc.data_ptr() + 4 * N, gg, N, 1, N)
        return dx, bag.result(bn.weight), bag.result(bn.bias), None, None, None


class MulFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f(a), _f(b)
        out = torch.empty_like(a)
        _hip.call("mm_mul_f32", a, b, out, a.numel())
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _f(g)
        da, db = torch.empty_like(a), torch.empty_like(b)
        _hip.call("mm_mul_f32", g, b, da, g.numel())
        _hip.call("mm_mul_f32", g, a, db, g.numel())
        return da, db


class Gate2MixFn(torch.autograd.Function):
    """HybridFusionModule mix: -> (comb (B, 2H), gate (B, 2))"""

    @staticmethod
    def forward(ctx, g, erp, pw, conn, boost):
        g, erp, pw, conn = _f(g), _f(erp), _f(pw), _f(conn)
        B, H = erp.shape
        comb = _empty((B, 2 * H), _F32, erp)
        gate = _empty((B, 2), _F32, erp)
        _hip.call("mm_gate2_mix", g, erp, pw, conn, comb, gate, B, H, float(boost))
        ctx.save_for_backward(g, erp, pw)
        ctx.boost = float(boost)
        ctx.mark_non_differentiable(gate)
        return comb, gate

    @staticmethod
    def backward(ctx, dcomb, _dgate):
        g, erp, pw = ctx.saved_tensors
        B, H = erp.shape
        derp, dpw, dconn = torch.empty_like(erp), torch.empty_like(erp), torch.empty_like(erp)
        dg = torch.empty_like(g)
        _hip.call("mm_gate2_mix_bwd", _f(dcomb), g, erp, pw, derp, dpw, dconn, dg, B, H, ctx.boost)
        return dg, derp, dpw, dconn, None


class LiteConvFn(torch.autograd.Function):
    """conv part of LiteERPEncoder / LitePowerEncoder (train mode):
    Conv-BN-GELU-Drop-MaxPool2 -> Conv-BN-GELU-Drop -> mean over time : (B,C,T) -> (B,H) fp32"""

    @staticmethod
    def forward(ctx, m, x, *params):
        cl = m.conv_layers
        p = m.drop_p
        xb = ops.pack_nct(_f(x))
        r1, s1 = ops.conv_bn_act(xb, cl[0], cl[1], pool=2, training=True, drop_p=p, drop_first=True, need_dgrad=False)
        r2, s2 = ops.conv_bn_act(r1["bf16"], cl[5], cl[6], training=True, drop_p=p, want_f32=True,
                                 want_bf16=False, need_dgrad=True)
        h = r2["f32"]
        B, T2, N = h.shape
        pooled = _empty((B, N), _F32, h)
        _hip.call("mm_meanpool_fwd", h, pooled, None, B, T2, N)
        ctx.saved, ctx.params, ctx.dims = (s1, s2), params, (B, T2, N, tuple(x.shape), x.requires_grad)
        return pooled

    @staticmethod
    def backward(ctx, dout):
        s1, s2 = ctx.saved
        B, T2, N, xshape, need_dx = ctx.dims
        dh = _empty((B, T2, N), _F32, dout)
        _hip.call("mm_meanpool_bwd", _f(dout), dh, B, T2, N)
        bag = GradBag()
        g = conv_bn_act_bwd(bag, s2, dout_f32=dh)
        g = conv_bn_act_bwd(bag, s1, dout_bf16=g, need_dx=need_dx)
        dx = None
        if need_dx:
            Bx, C, T = xshape
            dx = _empty((Bx, C, T), _F32, dout)
            _hip.call("mm_unpack_ntc_f32", g, dx, Bx, C, T, g.shape[2])
        return (None, dx) + tuple(bag.result(p) for p in ctx.params)


class SmoothedCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, smoothing):
        logits = _f(logits)
        B, C = logits.shape
        out = _zeros((1,), logits)
        dl = _empty((B, C), _F32, logits)
        _hip.call("mm_smoothed_ce", logits, target.contiguous(), out, dl, B, C, float(smoothing))
        ctx.save_for_backward(dl)
        return out.reshape(()).clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


# -------------------------------------------------------------- compositions
def linear(x, lin, act="none", drop_p=0.0):
    return SmallLinearFn.apply(x, lin.weight, lin.bias, act, float(drop_p))


def linear_bn_act(x, lin, bn, act, drop_p):
    """Linear -> BatchNorm1d(train) -> act -> Dropout"""
    return BNRowsActFn.apply(linear(x, lin), bn.weight, bn.bias, bn, act, float(drop_p))

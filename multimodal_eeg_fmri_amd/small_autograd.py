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
    """y = dropout(act(BatchNorm1d(x))) on fp32 (B, N).  Train mode: batch statistics, running stats updated.
    ``frozen`` (eval mode with a backward to follow - saliency / fine-tuning through a classifier head,
    crossmodal_v4_enhancements.py:909-915): running statistics, no update, no dropout; the backward is
    dx = scale * dz with the parameter gradients dbeta = sum dz, dgamma = sum dz * xhat."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, act, drop_p, frozen=False):
        x = _f(x)
        B, N = x.shape
        if frozen:
            out4 = ops.bn_fold_eval(bn, None)
            drop_p = 0.0
        else:
            stats = _zeros((REPL, 2, N), x)
            _hip.call("mm_colstats", x, stats, B, N)                 # replica 0
            out4 = ops.bn_finalize_train(bn, stats, B)
        ctx.frozen = bool(frozen)
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
        _hip.call("mm_bn_act_bwd_apply", x, out4, None, dy, sc, None, dx, *args, 0 if ctx.frozen else 1, 1)
        bag = GradBag()
        gg, gb = bag.target(bn.weight), bag.target(bn.bias)
        if gb is not None:
            _hip.call("mm_reduce_replicas", sc, gb, N, 1, N)
        if gg is not None:
            _hip.call("mm_reduce_replicas", sc.data_ptr() + 4 * N, gg, N, 1, N)
        return dx, bag.result(bn.weight), bag.result(bn.bias), None, None, None, None


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


def linear_bn_act(x, lin, bn, act, drop_p, frozen: bool = False):
    """Linear -> BatchNorm1d (train: batch statistics; ``frozen``: running statistics, differentiable) -> act -> Dropout"""
    return BNRowsActFn.apply(linear(x, lin), bn.weight, bn.bias, bn, act, float(drop_p), bool(frozen))


class LayerNormFn(torch.autograd.Function):
    """fp32 LayerNorm over the last dim of (B, D) rows."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _f(x)
        B, D = x.shape
        y = _empty((B, D), _F32, x)
        stat = _empty((B, 2), _F32, x)
        _hip.call("mm_layernorm_fwd", x, gamma, beta, None, y, stat, B, D, float(eps))
        ctx.save_for_backward(x, stat)
        ctx.g, ctx.b = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stat = ctx.saved_tensors
        B, D = x.shape
        dx = _empty((B, D), _F32, x)
        dgb = _zeros((REPL, 2, D), x)
        _hip.call("mm_layernorm_bwd", None, _f(dy), x, stat, ctx.g, None, dx, None, dgb, B, D, 0.0, 0, None)
        bag = GradBag()
        gw, gb = bag.target(ctx.g), bag.target(ctx.b)
        if gw is not None:
            _hip.call("mm_acc_reduce", dgb, gw, D, 2 * D)
        if gb is not None:
            _hip.call("mm_acc_reduce", dgb.data_ptr() + 8 * D, gb, D, 2 * D)
        return dx, bag.result(ctx.g), bag.result(ctx.b), None


class ActFn(torch.autograd.Function):
    """y = dropout(act(z)) elementwise, fp32."""

    @staticmethod
    def forward(ctx, z, act, drop_p):
        z = _f(z)
        seed = ops._next_seed() if drop_p > 0 else 0
        y = torch.empty_like(z)
        _hip.call("mm_act_f32", z, y, z.numel(), ACT[act], float(drop_p), seed, ops.EP())
        ctx.save_for_backward(z)
        ctx.meta = (act, float(drop_p), seed)
        return y

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        act, p, seed = ctx.meta
        out = torch.empty_like(z)
        _hip.call("mm_act_bwd_f32", _f(g), z, out, z.numel(), ACT[act], p, seed, ops.EP())
        return out, None, None


class Attn1x2Fn(torch.autograd.Function):
    """bridge cross-attention core: (proj_e, proj_f) (B, 3E) -> ctx (B, E), attw (B, 2)"""

    @staticmethod
    def forward(ctx, pe, pf, nhead, drop_p):
        pe, pf = _f(pe), _f(pf)
        B, E3 = pe.shape
        E = E3 // 3
        seed = ops._next_seed() if drop_p > 0 else 0
        out = _empty((B, E), _F32, pe)
        attw = _empty((B, 2), _F32, pe)
        _hip.call("mm_attn_1x2_train", pe, pf, None, out, attw, None, None, B, E, int(nhead), float(drop_p), seed,
                  ops.EP(), 0)
        ctx.save_for_backward(pe, pf)
        ctx.meta = (int(nhead), float(drop_p), seed)
        ctx.mark_non_differentiable(attw)
        return out, attw

    @staticmethod
    def backward(ctx, dctx, _dattw):
        pe, pf = ctx.saved_tensors
        nhead, p, seed = ctx.meta
        B, E3 = pe.shape
        dpe, dpf = torch.empty_like(pe), torch.empty_like(pf)
        _hip.call("mm_attn_1x2_train", pe, pf, _f(dctx), None, None, dpe, dpf, B, E3 // 3, nhead, p, seed, ops.EP(), 1)
        return dpe, dpf, None, None


class Attn1xKFn(torch.autograd.Function):
    """MultiheadAttention core, one query token x K <= 4 key/value tokens: (nhead, p, *in-projected
    tokens (B, 3E)) -> (context (B, E), head-averaged weights (B, K))."""

    @staticmethod
    def forward(ctx, nhead, drop_p, *projs):
        projs = [_f(p) for p in projs]
        K = len(projs)
        B, E3 = projs[0].shape
        E = E3 // 3
        seed = ops._next_seed() if drop_p > 0 else 0
        out = _empty((B, E), _F32, projs[0])
        attw = _empty((B, K), _F32, projs[0])
        pads = projs + [None] * (4 - K)
        _hip.call("mm_attn_1xk", *pads, K, None, out, attw, None, None, None, None, B, E, int(nhead),
                  float(drop_p), seed, ops.EP(), 0)
        ctx.save_for_backward(*projs)
        ctx.meta = (int(nhead), float(drop_p), seed, K, B, E)
        ctx.mark_non_differentiable(attw)
        return out, attw

    @staticmethod
    def backward(ctx, dctx, _dattw):
        projs = list(ctx.saved_tensors)
        nhead, p, seed, K, B, E = ctx.meta
        dps = [torch.empty_like(t) for t in projs]
        _hip.call("mm_attn_1xk", *(projs + [None] * (4 - K)), K, _f(dctx), None, None, *(dps + [None] * (4 - K)),
                  B, E, nhead, p, seed, ops.EP(), 1)
        return (None, None) + tuple(dps)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f(a), _f(b)
        out = torch.empty_like(a)
        _hip.call("mm_add_f32", a, b, out, a.numel())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def mha_1xk(attn, tokens, training: bool):
    """nn.MultiheadAttention(batch_first) with query = tokens[0] (one token) over keys/values = tokens:
    -> (output (B, E), head-averaged attention weights (B, K))"""
    projs = [SmallLinearFn.apply(t, attn.in_proj_weight, attn.in_proj_bias, "none", 0.0) for t in tokens]
    c, w = Attn1xKFn.apply(attn.num_heads, float(attn.dropout) if training else 0.0, *projs)
    return linear(c, attn.out_proj), w


class LearnedFusionFn(torch.autograd.Function):
    """LearnedFusionModule combine: (dyn (B, M), logits (M), temperature (), f_0..f_{M-1}) -> (fused, w)"""

    @staticmethod
    def forward(ctx, dyn, logits, temp, *feats):
        feats = [_f(f) for f in feats]
        M = len(feats)
        B, H = feats[0].shape
        dyn = _f(dyn)
        fused = _empty((B, H), _F32, dyn)
        w = _empty((B, M), _F32, dyn)
        tl = temp.detach().reshape(1).float().contiguous()
        ll = logits.detach().float().contiguous()
        _hip.call("mm_learned_fusion", feats[0], feats[1], feats[2] if M > 2 else None, dyn, ll, tl, fused, w, B, H, M)
        ctx.save_for_backward(dyn, ll, tl, *feats)
        ctx.params = (logits, temp)
        ctx.mark_non_differentiable(w)
        return fused, w

    @staticmethod
    def backward(ctx, dfused, _dw):
        dyn, ll, tl, *feats = ctx.saved_tensors
        logits, temp = ctx.params
        M = len(feats)
        B, H = feats[0].shape
        dfs = [torch.empty_like(f) for f in feats]
        ddyn = torch.empty_like(dyn)
        dl = _zeros((M,), dyn)
        dt = _zeros((1,), dyn)
        _hip.call("mm_learned_fusion_bwd", feats[0], feats[1], feats[2] if M > 2 else None, dyn, ll, tl, _f(dfused),
                  dfs[0], dfs[1], dfs[2] if M > 2 else None, ddyn, dl, dt, B, H, M)
        dtemp = dt.reshape(()) if temp.requires_grad else None
        return (ddyn, dl if logits.requires_grad else None, dtemp) + tuple(dfs)


class Softmax2ConcatFn(torch.autograd.Function):
    """fMRIFusionNet weighted concat: [softmax(pa,pc)_0 * a | softmax_1 * c]"""

    @staticmethod
    def forward(ctx, a, c, pa, pc):
        a, c = _f(a), _f(c)
        B, Ha = a.shape
        Hc = c.shape[1]
        out = _empty((B, Ha + Hc), _F32, a)
        pa_, pc_ = pa.detach().float().contiguous(), pc.detach().float().contiguous()
        _hip.call("mm_softmax2_concat", a, c, pa_, pc_, out, B, Ha, Hc)
        ctx.save_for_backward(a, c, pa_, pc_)
        return out

    @staticmethod
    def backward(ctx, g):
        a, c, pa_, pc_ = ctx.saved_tensors
        B, Ha = a.shape
        Hc = c.shape[1]
        da, dc = torch.empty_like(a), torch.empty_like(c)
        dpa, dpc = _zeros((1,), a), _zeros((1,), a)
        _hip.call("mm_softmax2_concat_bwd", _f(g), a, c, pa_, pc_, da, dc, dpa, dpc, B, Ha, Hc)
        return da, dc, dpa, dpc


class WeightedCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight):
        logits = _f(logits)
        B, C = logits.shape
        out = _zeros((1,), logits)
        dl = _empty((B, C), _F32, logits)
        _hip.call("mm_weighted_ce", logits, target.contiguous(), weight, out, dl, B, C)
        ctx.save_for_backward(dl)
        return out.reshape(()).clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


class DropPathFn(torch.autograd.Function):
    """per-sample stochastic depth; forward and backward are the same masked scale (``mm_drop_path``)."""

    @staticmethod
    def forward(ctx, x, drop_p):
        x = _f(x)
        seed = ops._next_seed()
        y = torch.empty_like(x)
        B = x.shape[0]
        _hip.call("mm_drop_path", x, y, B, x.numel() // B, drop_p, seed, ops.EP())
        ctx.meta = (drop_p, seed)
        return y

    @staticmethod
    def backward(ctx, g):
        drop_p, seed = ctx.meta
        g = _f(g)
        out = torch.empty_like(g)
        B = g.shape[0]
        _hip.call("mm_drop_path", g, out, B, g.numel() // B, drop_p, seed, ops.EP())
        return out, None


class FocalLossFn(torch.autograd.Function):
    """FocalLoss(alpha, gamma, reduction) of the EEG notebook (cell 20) in one launch (``mm_focal_loss``)."""

    @staticmethod
    def forward(ctx, logits, target, alpha, gamma, reduction):
        logits = _f(logits)
        B, C = logits.shape
        out = _zeros((1,), logits)
        per = _empty((B,), _F32, logits) if reduction == "none" else None
        dl = _empty((B, C), _F32, logits)
        scale = 1.0 / B if reduction == "mean" else 1.0
        _hip.call("mm_focal_loss", logits, target.contiguous(), out, per, dl, B, C, float(alpha), float(gamma), scale)
        ctx.save_for_backward(dl)
        ctx.factor = scale
        ctx.reduction = reduction
        return per if reduction == "none" else out.reshape(()).clone()

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if ctx.reduction == "none":
            return dl * g.reshape(-1, 1), None, None, None, None
        return dl * (g * ctx.factor), None, None, None, None


def proj_head(x, seq, drop_p):
    """Linear -> LayerNorm -> GELU -> Dropout (bridge_utils.py:34-45)"""
    return ActFn.apply(LayerNormFn.apply(linear(x, seq[0]), seq[1].weight, seq[1].bias, seq[1].eps), "gelu", float(drop_p))

"""Backward passes of the HIP path, exposed to torch.autograd as a handful of
coarse ``Function``s (one per encoder / block / head) so the tape stays short.

Parameter gradients are produced without floating-point atomics (per-workgroup slots summed in
order, or fixed-point accumulator workspaces: csrc/common.h), so a step is bit-reproducible.  When a
parameter carries a gradient *sink* (``param._mm_grad``, a view into the
trainer's flat gradient bucket) the kernels accumulate straight into it and
autograd receives ``None``; otherwise a fresh tensor is returned to autograd.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import _hip
from . import ops
from .ops import ACT, AREPL, REPL, WREP, _BF, _F32, _empty, _zeros


_BAG = {"cur": None}


def _reduce_into(dst, src, K, stride, offset=0, nrep=REPL):
    """dst[k] += sum_rep src[rep*stride + offset + k]   (deferred to GradBag.flush when a bag is active).
    nrep = REPL: ``src`` is a gradient accumulator workspace (zeroed (REPL, ...) fp32-sized buffer holding AREPL
    replicas of 64-bit fixed-point sums; ``stride`` / ``offset`` count its 64-bit elements); nrep = 1: compact fp32."""
    if dst is None:
        return
    if nrep not in (1, REPL):
        raise ValueError("_reduce_into: nrep is 1 (fp32 vector) or REPL (accumulator workspace)")
    bag = _BAG["cur"]
    ptr = src.data_ptr() + (8 if nrep == REPL else 4) * offset
    if bag is not None:
        bag.defer(ptr, dst, K, AREPL if nrep == REPL else 1, stride, keep=src)
    elif nrep == REPL:
        _hip.call("mm_acc_reduce", ptr, dst, K, stride)
    else:
        _hip.call("mm_reduce_replicas", ptr, dst, K, 1, stride)


def _scatter_into(dw, ws, cout, cin, taps, cinp, nrep, window=None):
    """dw[n][c][tap] += sum_rep ws[rep][n][tap][c]   (deferred to GradBag.flush when a bag is active).
    ``window`` = (first output channel, first tap, workspace taps, workspace output channels): ``dw`` (cout, cin, taps) takes
    only that block of a wider workspace (the branches of EnhancedPowerEncoder's merged convolution)."""
    bag = _BAG["cur"]
    if bag is not None:
        bag.defer_scatter(ws, dw, cout, cin, taps, cinp, nrep, window)
    elif window is None:
        _hip.call("mm_wgrad_scatter", ws, dw, cout, cin, taps, cinp, nrep)
    else:
        import ctypes
        import struct
        raw = struct.pack("<QQiiiiii", *scatter_desc(ws, dw, cout, cin, taps, cinp, nrep, window))
        host = ctypes.create_string_buffer(raw, len(raw))
        _hip.call("mm_scatter_many", ctypes.addressof(host), 1)


def scatter_desc(ws, dw, cout, cin, taps, cinp, nrep, window=None):
    """the 40-byte descriptor of mm_scatter_many / mm_flush_many (include/mmeeg_hip.h)"""
    if window is None:
        return (ws.data_ptr(), dw.data_ptr(), cout, cin, taps, cinp, nrep, 0)
    n0, tap0, ws_taps, ws_cout = window
    return (ws.data_ptr() + 4 * n0 * ws_taps * cinp, dw.data_ptr(), cout, cin, taps | (tap0 << 8) | (ws_taps << 16), cinp, nrep, ws_cout)


_SLOTS = {}


def _wgrad_slots(dy, x, dw, dbr, B, T, cinp, N, k, pad, cin, parts=None):
    """conv / linear weight gradient without atomics: every row-chunk workgroup stores its partial
    dW[n][tap][c] into its own slot of a workspace; the (deferred, batched) scatter sums the slots
    into the parameter layout."""
    import ctypes
    bag = _BAG["cur"]
    grouped = bag is not None and k == 1
    key = (B, T, cinp, N, k, grouped)
    slots = _SLOTS.get(key)
    if slots is None:
        out = ctypes.c_int(0)
        if grouped:
            _hip.call("mm_conv1d_wgrad_many_slots", B, T, cinp, N, ctypes.addressof(out))
        else:
            _hip.call("mm_conv1d_wgrad_slots", B, T, cinp, N, k, ctypes.addressof(out))
        slots = _SLOTS[key] = int(out.value)
    ws = _empty((slots, N, k, cinp), _F32, dy)                   # every element has exactly one writer: no memset
    if grouped:
        # a Linear's weight gradient feeds nothing but the final slot sum: collect it; the bag issues
        # all of them as ONE launch when it is flushed (possibly on another stream, off the chain)
        bag.defer_wgrad(dy, x, ws, dbr, B, T, cinp, N, slots)
    elif bag is not None and bag.defer_conv_wgrads:
        # k > 1 convs: same idea, one launch each at flush time (a trainer hands them to an idle stream)
        bag.calls.append(("mm_conv1d_wgrad", (dy, x, ws, dbr, B, T, cinp, N, k, pad, cinp, k * cinp, 1, cinp,
                                              slots, N * k * cinp, 1)))
    else:
        _hip.call("mm_conv1d_wgrad", dy, x, ws, dbr, B, T, cinp, N, k, pad, cinp, k * cinp, 1, cinp,
                  slots, N * k * cinp, 1)
    if parts is None:
        _scatter_into(dw, ws, N, cin, k, cinp, slots)
    else:            # (gradient tensor, first output channel, output channels, kernel size): centred tap windows of the workspace
        for t, n0, cn, kk in parts:
            if t is not None:
                _scatter_into(t, ws, cn, cin, kk, cinp, slots, window=(n0, (k - kk) // 2, k, N))


class deferred:
    """context: parameter-gradient replica reductions issued inside are collected
    in ``bag`` and executed as one mm_reduce_many launch on exit."""

    def __init__(self, bag, device):
        self.bag, self.device = bag, device

    def __enter__(self):
        self.prev = _BAG["cur"]
        _BAG["cur"] = self.bag
        return self.bag

    def __exit__(self, *exc):
        _BAG["cur"] = self.prev
        if exc[0] is None:
            self.bag.flush(self.device)
        return False


def _ln_param_grads(bag, ln, dgb, D):
    """dgb = replicated [REPL][2][D] {dgamma row, dbeta row}"""
    gw, gb = bag.target(ln.weight), bag.target(ln.bias)
    if gw is not None and gb is not None and gw.data_ptr() + 4 * D == gb.data_ptr():
        _reduce_into(gw, dgb, 2 * D, 2 * D)                                # adjacent in the flat bucket
        return
    _reduce_into(gw, dgb, D, 2 * D, 0)
    _reduce_into(gb, dgb, D, 2 * D, D)


def _compact(rep_buf, K):
    """gradient accumulator workspace of K values per replica -> compact fp32 [K] (one parallel reduction)"""
    out = _zeros((K,), rep_buf)
    _hip.call("mm_acc_reduce", rep_buf, out, K, K)
    return out


def _bn_param_grads(bag, bn, sums, N, nrep=1):
    """sums = [nrep][sum dz (dbeta) | sum dz*xhat (dgamma)]"""
    _reduce_into(bag.target(bn.bias), sums, N, 2 * N, 0, nrep=nrep)
    _reduce_into(bag.target(bn.weight), sums, N, 2 * N, N, nrep=nrep)


# ------------------------------------------------------------ gradient sinks
class GradBag:
    """collects parameter gradients of one backward call.  Replica reductions
    into parameter gradients are deferred and flushed as ONE kernel launch."""

    def __init__(self):
        self.fresh: Dict[int, torch.Tensor] = {}
        self.pending = []            # (src_ptr, dst_ptr, K, nrep, stride)
        self.scatters = []           # (ws_ptr, dw_ptr, Cout, Cin, taps, Cinp, nrep)
        self.wgrads = []             # (dy_ptr, x_ptr, ws_ptr, dbias_ptr, B, T, Cin, Cout, Cin_real, nslots)
        self.calls = []              # (entry point, args): launches postponed to flush time
        self.defer_conv_wgrads = False
        self._keep = []

    def defer(self, src_ptr: int, dst: torch.Tensor, K: int, nrep: int, stride: int, keep=None):
        self.pending.append((src_ptr, dst.data_ptr(), K, nrep, stride))
        self._keep.append((dst, keep))

    def defer_scatter(self, ws: torch.Tensor, dw: torch.Tensor, cout, cin, taps, cinp, nrep, window=None):
        self.scatters.append(scatter_desc(ws, dw, cout, cin, taps, cinp, nrep, window))
        self._keep.append((dw, ws))

    def defer_wgrad(self, dy, x, ws, dbr, B, T, cinp, N, slots):
        self.wgrads.append((dy.data_ptr(), x.data_ptr(), ws.data_ptr(), dbr.data_ptr() if dbr is not None else 0,
                            B, T, cinp, N, cinp, slots))
        self._keep.append((dy, x, ws, dbr))

    def hand_over(self) -> "GradBag":
        """move everything deferred so far into a new bag (to be flushed elsewhere, e.g. on another stream)"""
        other = GradBag()
        other.pending, other.scatters, other.wgrads, other._keep = self.pending, self.scatters, self.wgrads, self._keep
        other.calls = self.calls
        self.pending, self.scatters, self.wgrads, self.calls, self._keep = [], [], [], [], []
        return other

    def flush(self, device):
        import ctypes
        import struct
        for name, args in self.calls:                    # first: the slot sums and bias reductions below read them
            _hip.call(name, *args)
        self._keep.append(self.calls)
        self.calls = []
        if self.wgrads:
            raw = b"".join(struct.pack("<QQQQiiiiiiii", *d, 0, 0) for d in self.wgrads)
            host = ctypes.create_string_buffer(raw, len(raw))
            _hip.call("mm_conv1d_wgrad_many", ctypes.addressof(host), len(self.wgrads))
            self.wgrads = []
        if not self.scatters and not self.pending:
            return
        # slot sums (weight gradients) and accumulator reductions (bias / norm-parameter gradients): one launch
        sraw = b"".join(struct.pack("<QQiiiiii", *d) for d in self.scatters)
        rraw = b"".join(struct.pack("<QQqqq", *d) for d in self.pending)
        shost = ctypes.create_string_buffer(sraw, max(len(sraw), 1))      # descriptors travel as kernel arguments
        rhost = ctypes.create_string_buffer(rraw, max(len(rraw), 1))
        _hip.call("mm_flush_many", ctypes.addressof(shost) if self.scatters else None, len(self.scatters),
                  ctypes.addressof(rhost) if self.pending else None, len(self.pending))
        self.scatters, self.pending = [], []

    def target(self, p: torch.Tensor) -> Optional[torch.Tensor]:
        """fp32 buffer (PyTorch layout of ``p``) the kernels accumulate into."""
        if p is None or not p.requires_grad:
            return None
        sink = getattr(p, "_mm_grad", None)
        if sink is not None:
            return sink
        t = self.fresh.get(id(p))
        if t is None:
            t = torch.zeros_like(p, dtype=_F32, memory_format=torch.contiguous_format)
            self.fresh[id(p)] = t
        return t

    def result(self, p: torch.Tensor) -> Optional[torch.Tensor]:
        return self.fresh.get(id(p))


def _mask_cast(g_f32=None, g_bf16=None, z=None, act="none", drop_p=0.0, seed=0):
    """bf16( g * dropout_mask * act'(z) )"""
    src = g_f32 if g_f32 is not None else g_bf16
    out = _empty(src.shape, _BF, src)
    _hip.call("mm_act_bwd", g_f32, g_bf16, z, out, src.numel(), ACT[act], float(drop_p), int(seed), ops.EP())
    return out


def linear_bwd(bag: GradBag, dy: torch.Tensor, x: torch.Tensor, weight, bias, *, need_dx=True,
               dx_f32=False, below=None):
    """``below`` = (z, act, drop_p, seed) of the layer that produced ``x``: the data-gradient
    GEMM then returns d(pre-activation z) = (dy W) * act'(z) * dropout_mask directly."""
    """dy (M, N) bf16, x (M, Kp) bf16 -> dx (M, Kp); accumulates dW, db."""
    M, N = dy.shape
    Kp = x.shape[1]
    K = weight.shape[1]
    dw = bag.target(weight)
    db = bag.target(bias)
    dbr = _zeros((REPL, N), dy) if db is not None else None
    if dw is not None:
        _wgrad_slots(dy, x, dw, dbr, 1, M, Kp, N, 1, 0, K)
    elif db is not None:
        _hip.call("mm_colsum", dy, None, dbr, M, N)
    if db is not None:
        _reduce_into(db, dbr, N, N)
    if not need_dx:
        return None
    _, wd, cinp, coutp = ops.weights.get(weight, True)
    if coutp != N:
        raise _hip.HipLibraryError(f"linear_bwd: dY width {N} != padded out width {coutp}")
    kw = {}
    if below is not None:
        z, act, p, seed = below
        kw = dict(gradz=z, gradz_act=act, drop_p=p, seed=seed)
    r = ops.igemm(dy.view(1, M, N), wd, 1, 0, cinp, out_f32=dx_f32, out_bf16=not dx_f32, **kw)
    return (r["f32"] if dx_f32 else r["bf16"]).view(M, cinp)


_NO_BCAST = bool(os.environ.get("MM_NO_BCAST"))     # A/B knob: the voxel head's gradient as a full (B, V, N) tensor
_NO_GEMM2 = bool(os.environ.get("MM_NO_GEMM2"))     # A/B knob: the out-projection's data gradient as its own launch
_NO_BNRED = bool(os.environ.get("MM_NO_BNRED"))     # A/B knob: the BatchNorm-backward reduce as its own launch


def _bn_bwd_args(s: dict, y):
    B, T, N = y.shape
    d2 = s.get("drop2", (0.0, 0))
    return (B, T, N, ACT[s["act"]], s["pool"], 1 if s["drop_first"] else 0, float(s["drop_p"]), int(s["seed"]),
            float(d2[0]), int(d2[1]), ops.EP())


def conv_bn_act_bwd(bag: GradBag, s: dict, dout_bf16=None, dout_f32=None, need_dx=True, sums=None, below=None):
    """backward of ops.conv_bn_act (train mode). returns dx (B, T, Cinp) bf16.

    ``sums``: this block's BatchNorm-backward sums when the producer of ``dout`` already formed them (skips the
    reduce launch).  ``below`` = the saved dict of the conv block whose output fed this one: the data-gradient GEMM
    then runs that block's reduce pass as its epilogue (mm_conv1d_dgrad_bn_reduce) and the call returns
    ``(dx, sums_below)``; ``sums_below`` is None when the shapes do not allow the fusion (the caller passes it on
    either way)."""
    conv, bn = s["conv"], s["bn"]
    y, out4, xb = s["y"], s["out4"], s["xb"]
    B, T, N = y.shape
    args = _bn_bwd_args(s, y)
    if sums is None:
        sums = _zeros((REPL, 2, N), y)
        _hip.call("mm_bn_act_bwd_reduce", y, out4, dout_bf16, dout_f32, sums, *args)
    dy = _empty((B, T, N), _BF, y)
    # the apply pass sums the 32 replicas itself (no compaction launch between the two passes)
    _hip.call("mm_bn_act_bwd_apply", y, out4, dout_bf16, dout_f32, sums, dy, None, *args,
              1 if s.get("train", True) else 0, REPL)
    _bn_param_grads(bag, bn, sums, N, nrep=REPL)
    k, pad = conv.kernel_size[0], conv.padding[0]
    cin = conv.in_channels
    # a convolution merged from several modules (ops._power_merged_train) sends its slot sums straight into THEIR gradients
    parts = getattr(conv, "parts", None)
    if parts is not None:
        parts = [(bag.target(w), n0, w.shape[0], w.shape[2]) for w, n0 in parts]
        dw = None
    else:
        dw = bag.target(conv.weight)
    if dw is not None or (parts is not None and any(t is not None for t, *_ in parts)):
        cinp_x = xb.shape[2]
        db = bag.target(conv.bias)
        dbr = _zeros((REPL, N), y) if db is not None else None
        _wgrad_slots(dy, xb, dw, dbr, B, T, cinp_x, N, k, pad, cin, parts=parts)
        if db is not None:
            _reduce_into(db, dbr, N, N)
    if not need_dx:
        return None if below is None else (None, None)
    _, wd, cinp, coutp = ops.weights.get(conv.weight, True)
    assert coutp == N
    if below is not None:
        yb = below["y"]
        d2 = below.get("drop2", (0.0, 0))
        if not _NO_BNRED and (k > 1 or cinp <= 64) and yb.shape[2] == cinp and yb.shape[1] == T * below["pool"] and float(d2[0]) == 0.0:
            dx = _empty((B, T, cinp), _BF, y)
            sums_b = _zeros((REPL, 2, cinp), y)
            _hip.call("mm_conv1d_dgrad_bn_reduce", dy, wd, B, T, N, cinp, k, k - 1 - pad, dx, yb, below["out4"], sums_b,
                      ACT[below["act"]], below["pool"], 1 if below["drop_first"] else 0, float(below["drop_p"]),
                      int(below["seed"]), ops.EP())
            return dx, sums_b
        return ops.igemm(dy, wd, k, k - 1 - pad, cinp)["bf16"], None
    return ops.igemm(dy, wd, k, k - 1 - pad, cinp)["bf16"]


def _linear_ln_bwd(bag, dy, h, lin, wb, x, stat, ln, dres, dx, dx_bf16, dgb, drop_p, seed, bn_below=None, gemm2=None,
                   res_rows=0):
    """backward of ``Linear(LayerNorm(x))``: weight / bias gradients of the Linear (``h`` = LN(x) bf16 is its
    input), then d x = LN_backward(dy W) + dres.  Width 128 with M % 32 == 0 (the transformer blocks) runs
    the data-gradient GEMM with the LayerNorm backward as its epilogue; anything else takes two launches."""
    weight, bias = (lin.weight, lin.bias) if lin is not None else wb
    M, D = x.shape
    if D == 128 and M % 32 == 0:
        linear_bwd(bag, dy, h, weight, bias, need_dx=False)
        _, wd, cinp, coutp = ops.weights.get(weight, True)
        if coutp != dy.shape[1] or cinp != D:
            raise _hip.HipLibraryError(f"linear_ln_bwd: dY width {dy.shape[1]} / LN width {D} != weight image {coutp} x {cinp}")
        if bn_below is not None and dx_bf16 is None and dx is not None and not _NO_BNRED:
            # dx = the fp32 d(out) of the conv block below the stack: its BatchNorm-backward sums in the same launch
            sb = bn_below["s"]
            yb = sb["y"]
            d2 = sb.get("drop2", (0.0, 0))
            if yb.shape[2] == 128 and sb["pool"] == 1 and yb.shape[0] * yb.shape[1] == M:
                sums_b = _zeros((REPL, 2, 128), yb)
                _hip.call("mm_linear_dgrad_ln_bwd_bn_reduce", dy, wd, M, coutp, x, stat, ln.weight, dres, dx, dgb, ops.EP(),
                          yb, sb["out4"], sums_b, ACT[sb["act"]], float(sb["drop_p"]), int(sb["seed"]), float(d2[0]), int(d2[1]))
                bn_below["sums"] = sums_b
                return
        if gemm2 is not None and dx_bf16 is not None and not _NO_GEMM2:
            # ``gemm2`` = (128 x 128 data-gradient image, holder list): the data gradient of the Linear under this
            # LayerNorm's skip path (the attention out-projection), computed from dx_bf16 before it leaves the workgroup
            w2, holder = gemm2
            do = _empty((M, 128), _BF, dy)
            _hip.call("mm_linear_dgrad_ln_bwd_gemm2", dy, wd, M, coutp, x, stat, ln.weight, dres, dx, dx_bf16, dgb,
                      drop_p, seed, ops.EP(), w2, do, int(res_rows))
            holder.append(do)
            return
        _hip.call("mm_linear_dgrad_ln_bwd", dy, wd, M, coutp, x, stat, ln.weight, dres, dx, dx_bf16, dgb,
                  drop_p, seed, ops.EP())
        return
    dh = linear_bwd(bag, dy, h, weight, bias)
    _hip.call("mm_layernorm_bwd", dh, None, x, stat, ln.weight, dres, dx, dx_bf16, dgb, M, D, drop_p, seed, ops.EP())


def transformer_block_bwd(bag: GradBag, s: dict, dx2: torch.Tensor, dy2=None, emit_for=None, bn_below=None):
    """dx2 fp32 (M, D) -> (dx0 fp32 (M, D), bf16(dx0 * mask) or None).

    ``dy2`` = bf16(dx2 * dropout_mask(p, s3)) if the producer of dx2 already emitted it;
    ``emit_for`` = (p, seed) of the consumer of dx0 (the FFN2 mask of the block below):
    the last LayerNorm-backward then also writes that masked bf16 operand, so no
    stand-alone cast/mask pass runs between GEMMs.
    ``bn_below`` = {"s": saved dict of the conv block whose output is this block's input}: the last launch also forms
    that block's BatchNorm-backward sums (returned as ``bn_below["sums"]`` when the shapes allow the fusion)."""
    blk = s["blk"]
    p = s["p"]
    s1, s2, s3 = s["seeds"]
    B, L = s["B"], s["L"]
    at = blk.self_attn
    res_rows = 0
    if isinstance(dx2, tuple):                          # (one fp32 row per sample (B, D), tokens per sample): pooled_head_bwd
        rows, per = dx2
        M, D = dy2.shape
        _, _, ci, co = ops.weights.get(at.out_proj.weight, True)
        if D == 128 and M % 32 == 0 and ci == 128 and co == 128 and not _NO_GEMM2:
            dx2, res_rows = rows, per                   # the LayerNorm-backward launch reads the row (res_rows)
        else:                                           # no consumer for the row form: materialise the token gradients
            dx2 = rows.view(-1, 1, D).expand(-1, per, D).reshape(M, D).contiguous()
    else:
        M, D = dx2.shape
    # FFN second linear:  x2 = x1 + drop(g W2^T + b2);  g = drop(act(z))
    if dy2 is None:
        dy2 = _mask_cast(g_f32=dx2, drop_p=p, seed=s3)
    dz = linear_bwd(bag, dy2, s["g"], blk.linear2.weight, blk.linear2.bias, below=(s["z"], blk._act, p, s2))
    dx1 = _empty((M, D), _F32, dx2)
    dyo = _empty((M, D), _BF, dx2)                      # bf16(dx1 * mask1): out-proj backward operand
    dgb = _zeros((REPL, 2, D), dx2)
    # attention output projection:  x1 = x0 + drop(o Wo^T + bo): at width 128 its data gradient is a second GEMM inside the
    # launch above it (FFN-1 data gradient + norm2 backward), fed by the masked rows before they leave the workgroup
    holder = []
    g2 = None
    if D == 128 and M % 32 == 0:
        _, wd_o, cinp_o, coutp_o = ops.weights.get(at.out_proj.weight, True)
        if cinp_o == 128 and coutp_o == 128:
            g2 = (wd_o, holder)
    _linear_ln_bwd(bag, dz, s["h2"], blk.linear1, None, s["x1"], s["st2"], blk.norm2, dx2, dx1, dyo, dgb,
                   float(p), int(s1), gemm2=g2, res_rows=res_rows)
    _ln_param_grads(bag, blk.norm2, dgb, D)
    if holder:
        do = holder[0]
        linear_bwd(bag, dyo, s["o"].view(M, D), at.out_proj.weight, at.out_proj.bias, need_dx=False)
    else:
        do = linear_bwd(bag, dyo, s["o"].view(M, D), at.out_proj.weight, at.out_proj.bias)
    dqkv = _empty((B, L, 3 * D), _BF, dx2)
    delta = _empty((B, blk.nhead, L), _F32, dx2)
    dh = D // blk.nhead
    pa, sa = s.get("attn_drop", (0.0, 0))
    _hip.call("mm_attn_bwd", s["qkv"], s["o"], do, s["lse"], dqkv, delta, B, L, blk.nhead, dh, float(dh) ** -0.5,
              float(pa), int(sa), ops.EP(), s.get("mask"), ops.attn_mask_per_head(s.get("mask"), B, blk.nhead, L))
    dx0 = _empty((M, D), _F32, dx2)
    emit = _empty((M, D), _BF, dx2) if emit_for is not None else None
    ep, es = emit_for if emit_for is not None else (0.0, 0)
    dgb = _zeros((REPL, 2, D), dx2)
    _linear_ln_bwd(bag, dqkv.view(M, 3 * D), s["h1"], None, (at.in_proj_weight, at.in_proj_bias), s["x"], s["st1"],
                   blk.norm1, dx1, dx0, emit, dgb, float(ep), int(es), bn_below=bn_below)
    _ln_param_grads(bag, blk.norm1, dgb, D)
    return dx0, emit


def pooled_head_bwd(bag: GradBag, s: dict, dout: torch.Tensor, emit_for=None, rows_only=False):
    """dout fp32 (B, H) -> d tokens fp32 (B, L, D); with ``emit_for`` = (p, seed) of the consumer also its
    dropout-masked bf16 copy, returned as a pair (fused head only, else None).
    ``rows_only`` (fused head): returns (d pooled SUM fp32 (B, D), 1 / L) instead - every token's gradient is that row
    times the scale, and a consumer that can take it in this form (mm_bn_act_bwd_*_bcast) saves the (B, L, D) tensor."""
    lin = s["lin"]
    if s.get("fused") and rows_only and emit_for is None and not _NO_BCAST:
        B, L, D = s["B"], s["L"], s["D"]
        N = lin.weight.shape[0]
        dout = dout.contiguous()
        dz = _empty((B, N), _BF, dout)
        dp = _empty((B, 1, D), _F32, dout)
        _hip.call("mm_pooled_head_bwd", dout, s["z"], lin.weight, dz, dp, None, B, 1, D, N, ACT[s["act"]],
                  float(s["drop_p"]), int(s["seed"]), 0.0, 0, ops.EP())
        linear_bwd(bag, dz, s["pooled"], lin.weight, lin.bias, need_dx=False)
        return dp.view(B, D), 1.0 / L
    if s.get("fused") and rows_only and emit_for is not None and not _NO_BCAST:
        # the chain's form: masked bf16 tokens for the consumer's GEMMs + the one fp32 row per sample for its skip path
        B, L, D = s["B"], s["L"], s["D"]
        N = lin.weight.shape[0]
        dout = dout.contiguous()
        dz = _empty((B, N), _BF, dout)
        rows = _empty((B, D), _F32, dout)
        emit = _empty((B, L, D), _BF, dout)
        _hip.call("mm_pooled_head_bwd_rows", dout, s["z"], lin.weight, dz, rows, emit, B, L, D, N, ACT[s["act"]],
                  float(s["drop_p"]), int(s["seed"]), float(emit_for[0]), int(emit_for[1]), ops.EP())
        linear_bwd(bag, dz, s["pooled"], lin.weight, lin.bias, need_dx=False)
        return (rows, L), emit
    if s.get("fused"):
        B, L, D = s["B"], s["L"], s["D"]
        N = lin.weight.shape[0]
        dout = dout.contiguous()
        dz = _empty((B, N), _BF, dout)
        dx = _empty((B, L, D), _F32, dout)
        emit = _empty((B, L, D), _BF, dout) if emit_for is not None else None
        ep, es = emit_for if emit_for is not None else (0.0, 0)
        _hip.call("mm_pooled_head_bwd", dout, s["z"], lin.weight, dz, dx, emit, B, L, D, N, ACT[s["act"]],
                  float(s["drop_p"]), int(s["seed"]), float(ep), int(es), ops.EP())
        linear_bwd(bag, dz, s["pooled"], lin.weight, lin.bias, need_dx=False)
        return (dx, emit) if emit_for is not None else dx
    if emit_for is not None:
        return pooled_head_bwd(bag, s, dout), None
    dz = _mask_cast(g_f32=dout.contiguous(), z=s["z"], act=s["act"], drop_p=s["drop_p"], seed=s["seed"])
    dpool = linear_bwd(bag, dz, s["pooled"], lin.weight, lin.bias, dx_f32=True)
    B, L, D = s["B"], s["L"], s["D"]
    dx = _empty((B, L, D), _F32, dout)
    _hip.call("mm_meanpool_bwd", dpool, dx, B, L, D)
    return dx


def _module_params(m) -> List[torch.nn.Parameter]:
    return [p for p in m.parameters()]


class _ModuleFn(torch.autograd.Function):
    """shared plumbing: forward(ctx, module, x, *params) with params passed only
    so that autograd tracks them."""

    @staticmethod
    def _finish(ctx, bag, params, dx):
        return (None, dx) + tuple(bag.result(p) for p in params)


def erp_encoder_bwd(bag: GradBag, sv: dict, dout: torch.Tensor, need_dx: bool = False, after_blocks=None,
                    after_conv2=None, after_conv3=None):
    """backward of ops._erp_forward_impl (train mode); dout fp32 (B, H).  ``after_blocks()`` is called
    once the transformer stack's backward has been issued, ``after_conv2()`` once the second conv block's
    has (a trainer hands the reductions / weight gradients collected so far to another stream there)."""
    blocks = sv["blocks"]
    dy2 = None
    if blocks:                                       # the head's backward also writes the top block's masked operand
        d, dy2 = pooled_head_bwd(bag, sv["head"], dout, emit_for=(blocks[-1]["p"], blocks[-1]["seeds"][2]), rows_only=True)
    else:
        d = pooled_head_bwd(bag, sv["head"], dout)
    if isinstance(d, tuple):                         # (row per sample, tokens per sample): the top block reads the row
        B, L, D = dy2.shape
    else:
        B, L, D = d.shape
        d = d.view(B * L, D)
    if dy2 is not None:
        dy2 = dy2.view(B * L, D)
    c3, c2, c1 = sv["convs"][2], sv["convs"][1], sv["convs"][0]
    bn3 = {"s": c3}
    for i in range(len(blocks) - 1, -1, -1):
        below = (blocks[i - 1]["p"], blocks[i - 1]["seeds"][2]) if i > 0 else None
        d, dy2 = transformer_block_bwd(bag, blocks[i], d, dy2=dy2, emit_for=below, bn_below=bn3 if i == 0 else None)
    if after_blocks is not None:
        after_blocks()
    g, sm = conv_bn_act_bwd(bag, c3, dout_f32=d.view(B, L, D), sums=bn3.get("sums"), below=c2)
    if after_conv3 is not None:
        after_conv3()
    g, sm = conv_bn_act_bwd(bag, c2, dout_bf16=g, sums=sm, below=c1)
    if after_conv2 is not None:
        after_conv2()
    g = conv_bn_act_bwd(bag, c1, dout_bf16=g, need_dx=need_dx, sums=sm)
    if not need_dx:
        return None
    Bx, C, T = sv["x_shape"]
    dx = _empty((Bx, C, T), _F32, dout)
    _hip.call("mm_unpack_ntc_f32", g, dx, Bx, C, T, g.shape[2])
    return dx


class ErpEncoderFn(_ModuleFn):
    """train mode, or eval mode with a backward to follow (frozen BatchNorm: running statistics,
    no dropout) - the latter serves fine-tuning on a frozen encoder and gradient saliency
    (bridge_utils.py:158-229)."""

    @staticmethod
    def run(m, x):
        return ErpEncoderFn.apply(m, x, *_module_params(m))

    @staticmethod
    def forward(ctx, m, x, *params):
        out, saved = ops._erp_forward_impl(m, x.float(), m.training, True, save=True)
        ctx.m, ctx.saved, ctx.params = m, saved, params
        ctx.need_dx = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        bag = GradBag()
        with deferred(bag, dout.device):
            dx = erp_encoder_bwd(bag, ctx.saved, dout, ctx.need_dx)
        return _ModuleFn._finish(ctx, bag, ctx.params, dx)


def power_encoder_bwd(bag: GradBag, sv: dict, dout: torch.Tensor, need_dx: bool = False):
    """backward of ops._power_forward_impl; returns (dx packed bf16 (B, T, Cp) or None, finish) where
    ``finish()`` must run AFTER the bag's deferred reductions were flushed: it slices the merged
    192-channel conv / BatchNorm gradients back into the three conv_scale modules."""
    d = pooled_head_bwd(bag, sv["head"], dout)
    B, L, D = d.shape
    d = d.view(B * L, D)
    blocks = sv["blocks"]
    dy2 = None
    for i in range(len(blocks) - 1, -1, -1):
        below = (blocks[i - 1]["p"], blocks[i - 1]["seeds"][2]) if i > 0 else None
        d, dy2 = transformer_block_bwd(bag, blocks[i], d, dy2=dy2, emit_for=below)
    g = conv_bn_act_bwd(bag, sv["convs"][1], dout_f32=d.view(B, L, D))
    g = conv_bn_act_bwd(bag, sv["convs"][0], dout_bf16=g, need_dx=need_dx)
    conv, bn, seqs = sv["merged"]

    def finish():
        # ONE launch adds the merged bias / BatchNorm gradients' slices into the real parameters' sinks (it was twelve sliced
        # add_); the weight gradients went straight from the slot workspace into the three real tensors (conv_bn_act_bwd)
        tg = lambda ps: [bag.target(p) for p in ps]       # noqa: E731
        none3 = [None] * 3
        ops.power_merge_call(2, (tg([sq[0].weight for sq in seqs]), tg([sq[0].bias for sq in seqs]), tg([sq[1].weight for sq in seqs]),
                                 tg([sq[1].bias for sq in seqs]), none3, none3),
                             [bag.result(conv.weight), bag.result(conv.bias), bag.result(bn.weight), bag.result(bn.bias), None, None],
                             cin=conv.in_channels, ks=[sq[0].kernel_size[0] for sq in seqs])
    return g, finish


class PowerEncoderFn(_ModuleFn):
    """EnhancedPowerEncoder (train, or eval with a backward to follow).  ``packed``: x is already the
    channels-last bf16 (B, T, Cp) image (STFT front-end output) and gets no gradient."""

    @staticmethod
    def run(m, x, packed=False):
        return PowerEncoderFn.apply(m, x, packed, *_module_params(m))

    @staticmethod
    def forward(ctx, m, x, packed, *params):
        xb = x if packed else ops.pack_nct(x.float())
        need_dx = bool(x.requires_grad)                  # packed: the gradient goes back as the packed bf16 image (StftFrontEndFn)
        out, saved = ops._power_forward_impl(m, xb, m.training, need_dx, save=True)
        ctx.saved, ctx.params = saved, params
        ctx.need_dx, ctx.x_shape, ctx.packed = need_dx, tuple(x.shape), bool(packed)
        return out

    @staticmethod
    def backward(ctx, dout):
        bag = GradBag()
        with deferred(bag, dout.device):
            g, finish = power_encoder_bwd(bag, ctx.saved, dout.contiguous(), ctx.need_dx)
        finish()
        dx = None
        if ctx.need_dx and ctx.packed:
            dx = g
        elif ctx.need_dx:
            Bx, C, T = ctx.x_shape
            dx = _empty((Bx, C, T), _F32, dout)
            _hip.call("mm_unpack_ntc_f32", g, dx, Bx, C, T, g.shape[2])
        return (None, dx, None) + tuple(bag.result(p) for p in ctx.params)


class StftFrontEndFn(torch.autograd.Function):
    """multi-scale STFT power front-end [+ per-sample z-score] as a differentiable step: raw EEG (B, C, T) fp32 ->
    channels-last bf16 spectra (B, frames, Cp).  Backward: the gradient w.r.t. the packed spectra -> z-score backward
    (fp32) -> per scale, the gradient of |DFT(window * frame)|^2 gathered back onto the samples (mm_stft_power_bwd) ->
    d / d raw EEG (saliency / integrated gradients on the config-#5 model)."""

    @staticmethod
    def forward(ctx, x, n_ffts, hop, normalize):
        xc = x.float().contiguous()
        out, spec = ops.stft_front_end(xc, n_ffts, hop, normalize, keep_spec=True)
        ctx.save_for_backward(xc, spec if spec is not None else out)
        ctx.cfg = (tuple(n_ffts), int(hop), bool(normalize))
        return out

    @staticmethod
    def backward(ctx, g):
        xc, spec = ctx.saved_tensors
        n_ffts, hop, normalize = ctx.cfg
        B, C, T = xc.shape
        frames, cp = g.shape[1], g.shape[2]
        widths = [C * (n // 2 + 1) for n in n_ffts]
        g = g.contiguous()
        if normalize:
            gp = _empty((B, frames, cp), _F32, g)
            _hip.call("mm_sample_zscore_bwd", spec, g, gp, B, frames, sum(widths), cp, 1e-8)
        else:
            gp = g.float()
        dx = torch.zeros_like(xc)
        off = 0
        for n, wdt in zip(n_ffts, widths):
            _hip.call("mm_stft_power_bwd", xc, gp, dx, B, C, T, int(n), hop, off, cp)
            off += wdt
        return dx, None, None, None


class AddPositionalFn(torch.autograd.Function):
    """stand-alone PositionalEncoding.forward (enhanced_models_v4.py:44-55): dropout(x + pe[:L]);
    backward = the same counter-hash mask on the incoming gradient."""

    @staticmethod
    def forward(ctx, x, pe, p):
        seed = ops._next_seed() if p > 0 else 0
        ctx.p, ctx.seed, ctx.dtype = p, seed, x.dtype
        return ops.add_positional_impl(x, pe, p, seed)

    @staticmethod
    def backward(ctx, dout):
        if ctx.p > 0:
            dx = ops.add_positional_impl(dout, None, ctx.p, ctx.seed, backward=True)
        else:
            dx = dout
        return dx.to(ctx.dtype), None, None


class TransformerBlockFn(_ModuleFn):
    @staticmethod
    def run(blk, x, mask=None):
        return TransformerBlockFn.apply(blk, x, mask, *_module_params(blk))

    @staticmethod
    def forward(ctx, blk, x, mask, *params):
        xf = x.float().contiguous()
        out, saved = ops.transformer_block_fwd(xf, blk, blk.training, True, save=True, mask=mask)
        ctx.blk, ctx.saved, ctx.params = blk, saved, params
        return out

    @staticmethod
    def backward(ctx, dout):
        bag = GradBag()
        B, L, D = dout.shape
        with deferred(bag, dout.device):
            dx, _ = transformer_block_bwd(bag, ctx.saved, dout.contiguous().view(B * L, D).float())
        return (None, dx.view(B, L, D), None) + tuple(bag.result(p) for p in ctx.params)


def conv3d_bn_act_bwd(bag: GradBag, s: dict, dout, need_dx=True):
    """backward of ops.conv3d_bn_act (train, or eval with frozen BatchNorm: ``s["train"]`` False). dout: bf16 pooled
    volume, or fp32 (B, V, N) for the un-pooled last stage.  returns dx bf16 (B,D,H,W,Cinp)."""
    conv, bn = s["conv"], s["bn"]
    y, out4, xv = s["y"], s["out4"], s["xv"]
    B, D, H, W, N = y.shape
    train = 1 if s.get("train", True) else 0
    sums = _zeros((REPL, 2, N), y)
    dy = _empty((B, D, H, W, N), _BF, y)
    gelu = ACT["gelu"]
    if s["pool"]:
        _hip.call("mm_pool3d_bn_act_bwd_reduce", s["ysel"], out4, dout, sums, B, D, H, W, N, gelu,
                  float(s["drop_p"]), int(s["seed"]), ops.EP())
        # the apply passes sum the workspace's replicas themselves (no compaction launch between the two passes)
        _hip.call("mm_pool3d_bn_act_bwd_apply", y, s["arg"], out4, dout, sums, dy, B, D, H, W, N, gelu,
                  float(s["drop_p"]), int(s["seed"]), ops.EP(), train, REPL)
    elif isinstance(dout, tuple):                       # (one fp32 row per sample, scale): pooled_head_bwd(rows_only=True)
        rows, scale = dout
        _hip.call("mm_bn_act_bwd_reduce_bcast", y, out4, rows, float(scale), sums, B, D * H * W, N, gelu,
                  float(s["drop_p"]), int(s["seed"]), ops.EP())
        _hip.call("mm_bn_act_bwd_apply_bcast", y, out4, rows, float(scale), sums, dy, B, D * H * W, N, gelu,
                  float(s["drop_p"]), int(s["seed"]), ops.EP(), train, REPL)
    else:
        args = (B, D * H * W, N, gelu, 1, 1, float(s["drop_p"]), int(s["seed"]), 0.0, 0, ops.EP())
        _hip.call("mm_bn_act_bwd_reduce", y, out4, None, dout, sums, *args)
        _hip.call("mm_bn_act_bwd_apply", y, out4, None, dout, sums, dy, None, *args, train, REPL)
    _bn_param_grads(bag, bn, sums, N, nrep=REPL)
    cin = conv.in_channels
    dw = bag.target(conv.weight)
    if dw is not None:
        cinp_x = xv.shape[4]
        db = bag.target(conv.bias)
        dbr = _zeros((REPL, N), y) if db is not None else None
        import ctypes
        key = ("3d", B, D, H, W, cinp_x, N)
        slots = _SLOTS.get(key)
        if slots is None:
            out = ctypes.c_int(0)
            _hip.call("mm_conv3d_wgrad_slots", B, D, H, W, cinp_x, N, ctypes.addressof(out))
            slots = _SLOTS[key] = int(out.value)
        ws = _empty((slots, N, 27, cinp_x), _F32, y)           # one writer per element: no atomics, no memset
        end = ops.kernel_timer.bracket(f"conv3d_wgrad_c{cinp_x}")     # (bench.py's roofline_family; off by default)
        _hip.call("mm_conv3d_wgrad", dy, xv, ws, dbr, B, D, H, W, cinp_x, N, cinp_x,
                  27 * cinp_x, 1, cinp_x, slots, N * 27 * cinp_x, 1)
        if end is not None:
            end.record()
        _scatter_into(dw, ws, N, cin, 27, cinp_x, slots)
        if db is not None:
            _reduce_into(db, dbr, N, N)
    if not need_dx:
        return None
    _, wd, cinp, coutp = ops.weights.get(conv.weight, True)
    assert coutp == N
    dx = _empty((B, D, H, W, cinp), _BF, y)
    end = ops.kernel_timer.bracket(f"conv3d_dgrad_c{N}")
    _hip.call("mm_conv3d_fwd", dy, wd, B, D, H, W, N, cinp, None, None, None, dx)
    if end is not None:
        end.record()
    return dx


def conv3d_l1_bwd(bag: GradBag, s: dict, dout: torch.Tensor):
    """backward of ops.conv3d_l1_bn_act (no input gradient: the volume is data; a caller that wants one runs
    layer 1 through conv3d_bn_act, ops._vol_forward_impl(need_dx=True))."""
    conv, bn, x = s["conv"], s["bn"], s["x"]
    B, _, D, H, W = x.shape
    train = 1 if s.get("train", True) else 0
    # one recompute pass: S1 / S2 and A1 = x^T dz; BatchNorm's backward is linear in S1, S2, and its two correction terms
    # come from the forward pass's compact Gram matrix (csrc/conv3d_l1.hip)
    sums = _zeros((REPL, 2, 32), x)
    dw = bag.target(conv.weight)
    if dw is None:                                   # frozen conv weight: the sums alone (BatchNorm gradients)
        _hip.call("mm_conv3d_l1", 2, x, s["wimg"], conv.bias, s["out4"], dout, None, sums, None, None, None,
                  B, D, H, W, train, float(s["drop_p"]), int(s["seed"]), ops.EP())
    else:
        a1 = _zeros((REPL, 27, 32), x)
        _hip.call("mm_conv3d_l1_bwd", x, s["wimg"], conv.bias, s["out4"], dout, sums, a1, s.get("gramc") if train else None,
                  dw, bag.target(conv.bias), B, D, H, W, train, float(s["drop_p"]), int(s["seed"]), ops.EP())
    _bn_param_grads(bag, bn, sums, 32, nrep=REPL)


def volume_encoder_bwd(bag: GradBag, sv: dict, dout: torch.Tensor):
    """backward of ops._vol_forward_impl (train mode, or eval with frozen BatchNorm); dout fp32 (B, out_dim).
    Returns d / d volume (fp32, the input's shape) when the forward was run with need_dx, else None."""
    last = sv["convs"][2]
    d = pooled_head_bwd(bag, sv["head"], dout, rows_only=not last["pool"])     # fp32 (B, V, N), or (row per sample, 1 / V)
    g = conv3d_bn_act_bwd(bag, last, d)
    g = conv3d_bn_act_bwd(bag, sv["convs"][1], g)
    if sv["convs"][0].get("l1"):
        conv3d_l1_bwd(bag, sv["convs"][0], g)
        return None
    need_dx = bool(sv.get("need_dx"))
    dxp = conv3d_bn_act_bwd(bag, sv["convs"][0], g, need_dx=need_dx)
    if not need_dx:
        return None
    Bx, C, D, H, W = sv["x_shape"]
    return dxp[..., :C].permute(0, 4, 1, 2, 3).float().contiguous()      # (B, D, H, W, Cp) bf16 -> (B, C, D, H, W) fp32


class VolumeEncoderFn(_ModuleFn):
    """train mode, or eval mode with a backward to follow (frozen BatchNorm: running statistics, no dropout, no
    statistic update) - fine-tuning on a frozen voxel encoder and gradient saliency / integrated gradients on an
    end-to-end voxel path (the protocol of bridge_utils.py:158-229)."""

    @staticmethod
    def run(m, x):
        return VolumeEncoderFn.apply(m, x, *_module_params(m))

    @staticmethod
    def forward(ctx, m, x, *params):
        ctx.need_dx = bool(x.requires_grad)
        out, saved = ops._vol_forward_impl(m, x.float(), m.training, True, save=True, need_dx=ctx.need_dx)
        ctx.saved, ctx.params = saved, params
        return out

    @staticmethod
    def backward(ctx, dout):
        bag = GradBag()
        with deferred(bag, dout.device):
            dx = volume_encoder_bwd(bag, ctx.saved, dout)
        return _ModuleFn._finish(ctx, bag, ctx.params, dx)


# ------------------------------------------------ projection bridge / loss
def proj_head_bwd(bag: GradBag, s: dict, da: torch.Tensor, need_dx=True):
    lin, ln = s["seq"][0], s["seq"][1]
    B, N = da.shape
    K = lin.weight.shape[1]
    dhn = _empty((B, N), _F32, da)
    _hip.call("mm_act_bwd_f32", da, s["hn"], dhn, B * N, ACT["gelu"], float(s["p"]), int(s["seed"]), ops.EP())
    dz1 = _empty((B, N), _F32, da)
    dgb = _zeros((REPL, 2, N), da)
    _hip.call("mm_layernorm_bwd", None, dhn, s["z1"], s["stat"], ln.weight, None, dz1, None, dgb, B, N, 0.0, 0, None)
    _ln_param_grads(bag, ln, dgb, N)
    dx = _empty((B, K), _F32, da) if need_dx else None
    _hip.call("mm_small_linear_bwd", dz1, s["x"], lin.weight, dx, bag.target(lin.weight), bag.target(lin.bias), B, K, N)
    return dx


def contrastive_embed_bwd(bag: GradBag, s: dict, dz: torch.Tensor, need=(True, True)):
    """dz (B, 2N) packed -> (d eeg_feat (B, eeg_dim), d fmri_feat (B, fmri_dim))"""
    B, N = s["B"], s["N"]
    br = s["bridge"]
    (le, lne), (lf, lnf) = br.eeg_proj[:2], br.fmri_proj[:2]
    xe, xf = s["xe"], s["xf"]
    dxe = _empty(xe.shape, _F32, dz) if need[0] else None
    dxf = _empty(xf.shape, _F32, dz) if need[1] else None
    t = bag.target
    _hip.call("mm_proj_heads_bwd", dz.contiguous(), s["z"], s["nrm"], s["hn"], s["z1"], s["stat"],
              xe, le.weight, lne.weight, xe.shape[1], xf, lf.weight, lnf.weight, xf.shape[1],
              dxe, t(le.weight), t(le.bias), t(lne.weight), t(lne.bias),
              dxf, t(lf.weight), t(lf.bias), t(lnf.weight), t(lnf.bias),
              B, N, float(s["p"]), int(s["seeds"][0]), int(s["seeds"][1]), ops.EP())
    return dxe, dxf


class ContrastiveEmbedFn(torch.autograd.Function):
    @staticmethod
    def run(bridge, eeg, fmri):
        ps = list(bridge.eeg_proj.parameters()) + list(bridge.fmri_proj.parameters())
        return ContrastiveEmbedFn.apply(bridge, eeg, fmri, *ps)

    @staticmethod
    def forward(ctx, bridge, eeg, fmri, *params):
        z, saved = ops.contrastive_embed_impl(bridge, eeg, fmri, bridge.training)
        ctx.saved, ctx.params = saved, params
        ctx.need = (eeg.requires_grad, fmri.requires_grad)
        return z

    @staticmethod
    def backward(ctx, dz):
        bag = GradBag()
        with deferred(bag, dz.device):
            dxe, dxf = contrastive_embed_bwd(bag, ctx.saved, dz.contiguous(), ctx.need)
        return (None, dxe, dxf) + tuple(bag.result(p) for p in ctx.params)


class ClipLossFn(torch.autograd.Function):
    """z (B, 2N) packed local embeddings -> (loss, acc_e2f, acc_f2e).  With a process group the columns are
    the all-gathered global batch; every rank evaluates all rows of it (``mm_clip_loss_own_rows``), so the
    backward hands out d (sum over ranks of their losses) / d z_local directly - what a reduce-scatter of the
    gathered gradients would deliver - and the usual gradient all-reduce of data parallelism completes it."""

    @staticmethod
    def forward(ctx, z, logit_scale, group):
        from . import dp
        z = z.contiguous().float()
        B, N2 = z.shape
        N = N2 // 2
        world, rank = dp.world_size(group), dp.rank(group)
        z_all = dp.gather_embeddings(z, group)
        need_grad = z.requires_grad or logit_scale.requires_grad
        scal = _empty((4,), _F32, z)
        dz = _empty((B, N2), _F32, z) if need_grad else None
        ws = _empty((6 * world * B,), _F32, z)
        ls = logit_scale.detach().reshape(1).float().contiguous()
        _hip.call("mm_clip_loss_own_rows", z_all, ls, scal, dz, ws, B, world * B, N, rank * B)
        if need_grad:
            ctx.save_for_backward(dz, scal)
        return scal[0].clone(), scal[1].clone(), scal[2].clone()

    @staticmethod
    def backward(ctx, g_loss, g_a, g_b):
        dz, scal = ctx.saved_tensors
        return dz * g_loss, (scal[3] * g_loss).reshape(()), None

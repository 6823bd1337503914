"""Host-side composition of the HIP kernels (``libmmeeg_hip.so``) into the
reference's layers.  PyTorch is used for device memory, streams and autograd
bookkeeping only; every FLOP of the path runs in the C-ABI kernels.

Conventions
-----------
* activations between kernels are channels-last: EEG ``(B, T, C)``, tokens
  ``(B, L, d)``; GEMM operands are bf16, the transformer residual stream, all
  statistics and all parameter gradients are fp32;
* a Linear layer is the ``taps == 1`` case of the 1-D implicit GEMM;
* dropout masks are a counter hash of (seed, element index) recomputed in the
  backward kernels, never stored.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import _hip

ACT = {"none": 0, "gelu": 1, "relu": 2, "tanh": 3, "sigmoid": 4}
_BF = torch.bfloat16
_F32 = torch.float32


# --------------------------------------------------------------------- helpers
def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _hip.HipLibraryError(
                "multimodal_eeg_fmri_amd runs on the MI355X HIP path only: got a CPU tensor "
                "(no CPU / eager-PyTorch fallback is provided)")
    _hip.load()


def cpad(c: int) -> int:
    """channel count as the MFMA K-chunking wants it: a multiple of 16."""
    c16 = max(16, (c + 15) // 16 * 16)
    return c16


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


REPL = 32          # fp32-sized copies to allocate (and zero) per accumulator workspace (csrc/common.h: MM_REPL) ...
AREPL = 16         # ... which the kernels use as 16 replicas of 64-bit fixed-point sums (MM_ACC_REPL): order-free adds
WREP = 8           # replicas of the conv weight-gradient workspaces
ACC_STAT, ACC_GRAD = 28, 40        # fixed-point fraction bits: activation statistics / gradient sums (csrc/common.h)


def acc_decode(ws: torch.Tensor, k: int) -> torch.Tensor:
    """accumulator workspace (REPL, ...) as the kernels left it -> its fp64 sums (...).  For tests and debugging:
    the product reads workspaces on the device (mm_bn_finalize, mm_acc_reduce, ...)."""
    i = ws.contiguous().flatten().view(torch.int64).view(AREPL, *ws.shape[1:])
    out = i.sum(0).double() * 2.0 ** -k
    # csrc/common.h's overflow contract: a poisoned replica (2^62) or replica magnitudes summing to 2^61 read as NaN
    bad = i.double().abs().sum(0) >= 2.0 ** 61
    return torch.where(bad, torch.full_like(out, float("nan")), out)


def acc_encode(values: torch.Tensor, k: int) -> torch.Tensor:
    """fp sums (...) -> an accumulator workspace (REPL, ...) holding them (replica 0), e.g. to feed mm_bn_finalize"""
    i = torch.zeros((AREPL,) + tuple(values.shape), dtype=torch.int64, device=values.device)
    i[0] = (values.double() * 2.0 ** k).round().long()
    return i.flatten().view(torch.float32).view((REPL,) + tuple(values.shape))


class _Arena:
    """one zeroed fp32 scratch buffer per training step: every accumulator the
    kernels add into (stats, sums, wgrad workspaces ...) is a slice of it, so a
    step pays ONE memset instead of ~40 tiny fill launches."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.active = False
        self.cleared = 0
        self.high = 0                     # largest offset any step has reached

    def begin(self, device, nfloats: int = 6 << 20, clear: Optional[int] = None, defer_zero: bool = False):
        """``clear``: zero only the first ``clear`` floats (a caller that knows its step's high-water
        mark; slices beyond it fall back to torch.zeros).  ``defer_zero``: the caller zeroes
        ``zero_range()`` itself before the first kernel that uses a slice (the trainer's weight-image launch does)."""
        if self.buf is None or self.buf.device != device or self.buf.numel() < nfloats:
            self.buf = torch.empty(nfloats, dtype=_F32, device=device)
        self.cleared = self.buf.numel() if clear is None else min(int(clear), self.buf.numel())
        self.cleared -= self.cleared % 4
        if not defer_zero:
            self.buf[:self.cleared].zero_()
        self.off = 0
        self.active = True

    def end(self):
        self.active = False

    def zero_range(self):
        """(tensor, floats) a ``begin(defer_zero=True)`` left for the caller to clear"""
        return self.buf, self.cleared

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        if not self.active or self.buf.device != device or self.off + n > self.cleared:
            return None
        out = self.buf[self.off:self.off + n].view(shape)
        self.off += (n + 63) // 64 * 64
        self.high = max(self.high, self.off)
        return out


arena = _Arena()


def _zeros(shape, like, dtype=_F32):
    if dtype == _F32:
        t = arena.take(tuple(shape), like.device)
        if t is not None:
            return t
    return torch.zeros(shape, dtype=dtype, device=like.device)


_NO_SPLITK = bool(os.environ.get("MM_NO_SPLITK"))
_NO_FFN1_FUSE = not os.environ.get("MM_FFN1_FUSE")        # the first FFN Linear inside the out-projection's launch: measured, not
                                                           # faster (2-byte column stores of 2 x 16 MB) - off unless MM_FFN1_FUSE=1
_NO_QKV_FUSE = bool(os.environ.get("MM_NO_QKV_FUSE"))     # A/B knob: the next block's QKV projection as its own launch
_seed_state = {"base": 0x1234567, "step": 0, "epoch": None}


def set_seed_epoch(t: Optional[torch.Tensor]):
    """device int32 word mixed into every dropout seed ON THE DEVICE; lets a
    hipGraph-captured step draw fresh masks on every replay (None = off)."""
    _seed_state["epoch"] = t


def EP():
    return _seed_state["epoch"]


def set_dropout_seed(seed: int):
    _seed_state["base"] = int(seed) & 0x7FFFFFFF
    _seed_state["step"] = 0


def _next_seed() -> int:
    _seed_state["step"] += 1
    return (_seed_state["base"] * 2654435761 + _seed_state["step"] * 40503) & 0xFFFFFFFF


class _KernelTimer:
    """optional HIP-event bracket around named kernel launches on the current
    stream (bench.py uses it for the roofline line; off by default)."""

    def __init__(self):
        self._ev = {}

    def reset(self, name):
        self._ev[name] = []

    def bracket(self, name):
        lst = self._ev.get(name)
        if lst is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lst.append((a, b))
        a.record()
        return b

    def all_ms(self, name):
        lst = self._ev.get(name) or []
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in lst]

    def mean_ms(self, name):
        v = self.all_ms(name)
        return sum(v) / len(v) if v else None


kernel_timer = _KernelTimer()


# ------------------------------------------------------------ weight images
class _WeightCache:
    """bf16 MFMA images of fp32 parameters, rebuilt when the parameter changes.

    forward image  ``wf [Cout][k][Cinp]``; data-gradient image
    ``wd [Cinp][k flipped][Coutp]`` (``mm_prep_conv_weight``)."""

    def __init__(self):
        self._gen = 0
        self._store: Dict[int, tuple] = {}
        self._rec = None                      # list of (owner, view shape | None, need_dgrad) while recording

    def invalidate(self):
        self._gen += 1

    # ---- batched preparation: record which images one step asks for, then rebuild them all in
    # ONE launch at the start of every later step (a graph replay otherwise carries ~40 five-
    # microsecond repack nodes on its critical path)
    def start_recording(self):
        self._rec = []

    def stop_recording(self):
        rec, self._rec = self._rec or [], None
        seen, out = {}, []
        for owner, shape, dg in rec:
            k = (id(owner), shape)
            if k in seen:
                out[seen[k]][2] = out[seen[k]][2] or dg
            else:
                seen[k] = len(out)
                out.append([owner, shape, dg])
        return [tuple(e) for e in out]

    @staticmethod
    def _as3d(w: torch.Tensor) -> torch.Tensor:
        wd3 = w.detach()
        if wd3.dim() == 2:
            wd3 = wd3.unsqueeze(-1)
        if wd3.dim() == 5:
            wd3 = wd3.reshape(wd3.shape[0], wd3.shape[1], -1)
        return wd3.contiguous()

    def prepare_all(self, recorded, zero=None):
        """rebuild every recorded image that is out of date with one mm_prep_many launch; ``zero`` = (fp32 tensor,
        floats): the same launch clears that range (the step's accumulator arena)"""
        import ctypes
        import struct
        raw, keep = [], []
        for owner, shape, need_dgrad in recorded:
            w = owner if shape is None else owner.view(shape)
            ver = (owner.data_ptr(), owner._version, self._gen, tuple(w.shape))
            hit = self._store.get(id(owner))
            if hit is not None and hit[5]() is owner and hit[0] == ver and (hit[2] is not None or not need_dgrad):
                continue
            wd3 = self._as3d(w)
            cout, cin, k3 = wd3.shape
            cinp, coutp = cpad(cin), cpad(cout)
            wf = _empty((cout, k3, cinp), _BF, w)
            wd = _empty((cinp, k3, coutp), _BF, w) if need_dgrad else None
            raw.append(struct.pack("<QQQiiiiii", wd3.data_ptr(), wf.data_ptr(), wd.data_ptr() if wd is not None else 0,
                                   cout, cin, k3, cinp, coutp if need_dgrad else 0, 0))
            keep.append(wd3)
            self._store[id(owner)] = (ver, wf, wd, cinp, coutp, weakref.ref(owner))
        if raw:
            buf = b"".join(raw)
            host = ctypes.create_string_buffer(buf, len(buf))
            if zero is not None:
                _hip.call("mm_prep_many_zero", ctypes.addressof(host), len(raw), zero[0], int(zero[1]))
            else:
                _hip.call("mm_prep_many", ctypes.addressof(host), len(raw))
        elif zero is not None:
            zero[0][:int(zero[1])].zero_()

    def seed(self, w: torch.Tensor, wf: torch.Tensor, wd, cinp: int, coutp: int):
        """hand in images that were built elsewhere (mm_power_merge mode 3); ``get(w, ...)`` then returns them as long as
        ``w`` is unchanged and no data-gradient image is asked for that was not handed in"""
        self._store[id(w)] = ((w.data_ptr(), w._version, self._gen, tuple(w.shape)), wf, wd, cinp, coutp, weakref.ref(w))

    def get(self, w: torch.Tensor, need_dgrad: bool, key=None):
        owner = w if key is None else key          # the nn.Parameter the image belongs to
        if self._rec is not None and not getattr(owner, "_mm_transient", False):
            self._rec.append((owner, None if key is None else tuple(w.shape), bool(need_dgrad)))
        k = id(owner)
        ver = (owner.data_ptr(), owner._version, self._gen, tuple(w.shape))
        hit = self._store.get(k)
        # the weak reference rules out a dead parameter whose id/address got reused
        if (hit is not None and hit[5]() is owner and hit[0] == ver
                and (hit[2] is not None or not need_dgrad)):
            return hit[1], hit[2], hit[3], hit[4]
        wd3 = self._as3d(w)
        cout, cin, k3 = wd3.shape
        cinp, coutp = cpad(cin), cpad(cout)
        wf = _empty((cout, k3, cinp), _BF, w)
        wd = _empty((cinp, k3, coutp), _BF, w) if need_dgrad else None
        _hip.call("mm_prep_conv_weight", wd3, wf, wd, cout, cin, k3, cinp, coutp if need_dgrad else 0)
        if len(self._store) > 4096:
            self._store = {i: e for i, e in self._store.items() if e[5]() is not None}
        self._store[k] = (ver, wf, wd, cinp, coutp, weakref.ref(owner))
        return wf, wd, cinp, coutp


weights = _WeightCache()


def weights_changed():
    """call after parameters were updated outside torch's version counter
    (the fused AdamW kernel writes through raw pointers)."""
    weights.invalidate()


# ----------------------------------------------------------- kernel wrappers
def pack_nct(x: torch.Tensor) -> torch.Tensor:
    """(B, C, T) fp32 -> (B, T, Cp) bf16 channels-last, zero-padded channels."""
    B, C, T = x.shape
    cp = cpad(C)
    y = _empty((B, T, cp), _BF, x)
    _hip.call("mm_pack_nct_bf16", x.contiguous(), y, B, C, T, cp)
    return y


def igemm(x: torch.Tensor, wf: torch.Tensor, taps: int, pad: int, cout: int, *,
          scale=None, shift=None, act="none", residual=None, pe=None, pool=1, stats=None,
          out_f32=False, out_bf16=True, out_pre=False, drop_p=0.0, seed=0, gradz=None, gradz_act="none"):
    """x (B, T, Cin) bf16 -> dict(out_f32, out_bf16, out_pre) of (B, T/pool, cout)."""
    B, T, cin = x.shape
    res = {}
    of = _empty((B, T // pool, cout), _F32, x) if out_f32 else None
    ob = _empty((B, T // pool, cout), _BF, x) if out_bf16 else None
    op = _empty((B, T, cout), _BF, x) if out_pre else None
    args = (x, wf, B, T, cin, cout, taps, pad, scale, shift, ACT[act], residual, pe,
            pool, stats, of, ob, op, float(drop_p), int(seed), EP(), gradz, ACT[gradz_act])
    nsplit, ws_floats = _splitk_plan(B, T, cin, cout, taps)
    if nsplit > 1:                # few output tiles, long reduction (config #5's merged conv): channel slices + an epilogue launch
        _hip.call("mm_conv1d_fwd_splitk", *args, _empty((ws_floats,), _F32, x), nsplit)
    else:
        _hip.call("mm_conv1d_fwd", *args)
    res["f32"], res["bf16"], res["pre"] = of, ob, op
    return res


_SPLITK_PLANS: Dict[tuple, tuple] = {}


def _splitk_plan(B: int, T: int, cin: int, cout: int, taps: int):
    """(slices, workspace floats) of mm_conv1d_fwd_splitk_plan, cached per shape; MM_NO_SPLITK=1 turns it off (A/B)"""
    if taps == 1 or _NO_SPLITK:
        return 1, 0
    key = (B, T, cin, cout, taps)
    plan = _SPLITK_PLANS.get(key)
    if plan is None:
        import ctypes
        n, ws = ctypes.c_int(0), ctypes.c_int64(0)
        _hip.call("mm_conv1d_fwd_splitk_plan", B, T, cin, cout, taps, ctypes.addressof(n), ctypes.addressof(ws))
        plan = _SPLITK_PLANS[key] = (n.value, ws.value)
    return plan


def linear_rows(x2d: torch.Tensor, weight: torch.Tensor, bias, *, act="none", residual=None,
                out_f32=False, out_bf16=True, out_pre=False, drop_p=0.0, seed=0, need_dgrad=False):
    """(M, K) bf16 @ weight(N, K)^T + bias with a fused epilogue (taps == 1)."""
    M, K = x2d.shape
    wf, _, cinp, _ = weights.get(weight, need_dgrad)
    if cinp != K:
        raise _hip.HipLibraryError(f"linear: activation width {K} != padded weight width {cinp}")
    r = igemm(x2d.view(1, M, K), wf, 1, 0, weight.shape[0], shift=bias, act=act,
              residual=residual, out_f32=out_f32, out_bf16=out_bf16, out_pre=out_pre,
              drop_p=drop_p, seed=seed)
    return {k: (v.view(-1, v.shape[-1]) if v is not None else None) for k, v in r.items()}


def bn_fold_eval(bn, conv_bias) -> torch.Tensor:
    """[4][N] = scale, shift (conv bias folded), mean, rstd from running stats."""
    n = bn.num_features
    out4 = _empty((4, n), _F32, bn.weight)
    _hip.call("mm_bn_finalize", None, bn.weight, bn.bias, bn.running_mean, bn.running_var,
              conv_bias, out4, n, 1.0, 0.0, float(bn.eps), 1, None)
    return out4


def bn_finalize_train(bn, stats, count) -> torch.Tensor:
    n = bn.num_features
    out4 = _empty((4, n), _F32, bn.weight)
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    _hip.call("mm_bn_finalize", stats, bn.weight, bn.bias, bn.running_mean, bn.running_var,
              None, out4, n, float(count), mom, float(bn.eps), 0, bn.num_batches_tracked)
    return out4


_NO_FIN_FOLD = bool(os.environ.get("MM_NO_FIN_FOLD"))      # A/B knob: BatchNorm finalize as its own launch (round 3)


def bn_fin_desc(bn, stats, count):
    """host-side mm_bn_fin_t for the `*_fin` apply passes (the train-mode finalize runs in their prologue) ->
    (ctypes buffer to pass as ``bn_fin_host``, out4 tensor the launch will write, objects to keep alive)"""
    import ctypes
    import struct
    n = bn.num_features
    out4 = _empty((4, n), _F32, bn.weight)
    mom = 0.1 if bn.momentum is None else float(bn.momentum)
    nb = bn.num_batches_tracked
    raw = struct.pack("<QQQQQQQfffi", stats.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                      bn.running_var.data_ptr(), out4.data_ptr(), nb.data_ptr() if nb is not None else 0,
                      float(count), mom, float(bn.eps), 0)
    buf = ctypes.create_string_buffer(raw, len(raw))
    return buf, out4


def layernorm(x2d: torch.Tensor, ln, want_stat: bool):
    M, D = x2d.shape
    out = _empty((M, D), _BF, x2d)
    stat = _empty((M, 2), _F32, x2d) if want_stat else None
    _hip.call("mm_layernorm_fwd", x2d, ln.weight, ln.bias, out, None, stat, M, D, float(ln.eps))
    return out, stat


def attn_effective_dropout(p: float) -> float:
    """the attention-probability dropout rate the kernels apply for a requested ``p``: csrc/attention.hip decides a 2 x 2 block
    of scores with one 32-bit hash and 8-bit fields, so the rate is quantised to round(256 p) / 256 (p = 0.1 -> 0.1016,
    the keep scale is that of the quantised rate: the mask stays unbiased); 0 < p < 1/512 rounds to NO dropout"""
    return round(256.0 * float(p)) / 256.0


def attention(qkv: torch.Tensor, nhead: int, want_lse: bool, drop_p: float = 0.0, seed: int = 0, mask=None):
    if 0.0 < float(drop_p) < 1.0 / 512.0:
        import warnings
        warnings.warn(f"attention dropout p = {drop_p} is below the kernels' resolution (1/256): no attention-probability dropout "
                      "is applied (nn.MultiheadAttention(dropout=p) would drop with probability p)")
    B, L, E3 = qkv.shape
    E = E3 // 3
    dh = E // nhead
    out = _empty((B, L, E), _BF, qkv)
    lse = _empty((B, nhead, L), _F32, qkv) if want_lse else None
    _hip.call("mm_attn_fwd", qkv, out, lse, B, L, nhead, dh, 1.0 / math.sqrt(dh), float(drop_p), int(seed), EP(), mask,
              attn_mask_per_head(mask, B, nhead, L))
    return out, lse


def attn_mask_per_head(mask, B: int, nhead: int, L: int) -> int:
    """0: ``mask`` is None or the shared (L, L) matrix; 1: the (B * nhead, L, L) form - its leading size is checked
    here, where the batch is known"""
    if mask is None or mask.dim() == 2:
        return 0
    if tuple(mask.shape) != (B * nhead, L, L):
        raise ValueError(f"attn_mask: the 3-D form must be (batch * heads, L, L) = ({B * nhead}, {L}, {L}), got {tuple(mask.shape)}")
    return 1


def additive_attn_mask(mask, L: int, like: torch.Tensor):
    """nn.MultiheadAttention's ``attn_mask`` (enhanced_models_v4.py:98) as the additive fp32 matrix the kernels take -
    (L, L) shared by all heads (the only form the reference passes) or (batch * heads, L, L): a boolean mask marks
    NOT-allowed positions with True (-> -inf)."""
    if mask is None:
        return None
    if mask.dim() not in (2, 3) or tuple(mask.shape[-2:]) != (L, L):
        raise ValueError(f"attn_mask: expected (L, L) or (batch * heads, L, L) with L = {L}, got {tuple(mask.shape)}")
    if mask.dtype == torch.bool:
        m = torch.zeros(tuple(mask.shape), dtype=_F32, device=like.device)
        m.masked_fill_(mask.to(like.device), float("-inf"))
        return m
    return mask.to(device=like.device, dtype=_F32).contiguous()


# ------------------------------------------------------------------- stages
def conv_bn_act(xb: torch.Tensor, conv, bn, *, act="gelu", pool=1, training=False, drop_p=0.0,
                drop_first=True, pe=None, pe_drop_p=0.0, want_f32=False, want_bf16=True, need_dgrad=False,
                save=None, next_norm=None):
    """Conv1d -> BatchNorm1d -> act [-> MaxPool(2)] [-> Dropout] on (B, T, Cp) bf16.

    eval : one kernel (BN folded into the GEMM epilogue).
    train: GEMM (+bias, per-channel sum/sumsq) -> finalize -> BN/act/pool apply.
    eval with ``save`` (a backward will follow: frozen BatchNorm, saliency maps): the
    train-shaped pipeline with the RUNNING statistics and no statistic update.
    ``next_norm`` (a LayerNorm(128) that consumes the fp32 output row-wise - the transformer stack's first norm1):
    computed by the same launch when the block is 128 wide and un-pooled; the result comes back as ``out["prenorm"]``
    = (bf16 rows, mean / rstd or None).
    Returns (out dict, saved-for-backward dict or None)."""
    save = training if save is None else save
    k = conv.kernel_size[0]
    pad = conv.padding[0]
    cout = conv.out_channels
    wf, _, cinp, _ = weights.get(conv.weight, need_dgrad)
    assert xb.shape[2] == cinp, (xb.shape, cinp)
    B, T, _ = xb.shape
    if not save:
        out4 = bn_fold_eval(bn, conv.bias)
        r = igemm(xb, wf, k, pad, cout, scale=out4[0], shift=out4[1], act=act, pe=pe, pool=pool,
                  out_f32=want_f32, out_bf16=want_bf16)
        return r, None
    stats = _zeros((REPL, 2, cout), xb) if training else None
    y = igemm(xb, wf, k, pad, cout, shift=conv.bias, stats=stats, out_f32=True, out_bf16=False)["f32"]
    # train mode, GELU, <= 256 channels: the finalize (mean / rstd / running statistics) runs in the apply pass's prologue
    fold = training and act == "gelu" and cout <= 256 and not _NO_FIN_FOLD and isinstance(bn.running_mean, torch.Tensor)
    fin = None
    if fold:
        fin, out4 = bn_fin_desc(bn, stats, B * T)
    else:
        out4 = bn_finalize_train(bn, stats, B * T) if training else bn_fold_eval(bn, None)
    if not training:
        drop_p = pe_drop_p = 0.0
    seed = _next_seed() if drop_p > 0 else 0
    seed2 = _next_seed() if pe_drop_p > 0 else 0
    of = _empty((B, T // pool, cout), _F32, xb) if want_f32 else None
    ob = _empty((B, T // pool, cout), _BF, xb) if want_bf16 else None
    prenorm = None
    if next_norm is not None and cout == 128 and pool == 1 and want_f32 and not want_bf16 and drop_first \
            and tuple(next_norm.normalized_shape) == (128,):
        hn = _empty((B * T, cout), _BF, xb)
        stn = _empty((B * T, 2), _F32, xb)
        if fold:
            import ctypes
            _hip.call("mm_bn_act_fwd_ln_fin", y, ctypes.addressof(fin), pe, of, B, T, ACT[act], float(drop_p), seed,
                      float(pe_drop_p), seed2, EP(), next_norm.weight, next_norm.bias, float(next_norm.eps), hn, stn)
        else:
            _hip.call("mm_bn_act_fwd_ln", y, out4[0], out4[1], pe, of, B, T, ACT[act], float(drop_p), seed,
                      float(pe_drop_p), seed2, EP(), next_norm.weight, next_norm.bias, float(next_norm.eps), hn, stn)
        prenorm = (hn, stn)
    elif fold:
        import ctypes
        _hip.call("mm_bn_act_fwd_fin", y, ctypes.addressof(fin), pe, ob, of, B, T, cout, ACT[act], pool,
                  1 if drop_first else 0, float(drop_p), seed, float(pe_drop_p), seed2, EP())
    else:
        _hip.call("mm_bn_act_fwd", y, out4[0], out4[1], pe, ob, of, B, T, cout, ACT[act], pool,
                  1 if drop_first else 0, float(drop_p), seed, float(pe_drop_p), seed2, EP())
    saved = dict(xb=xb, y=y, out4=out4, act=act, pool=pool, drop_p=drop_p, seed=seed,
                 drop2=(float(pe_drop_p), seed2), drop_first=drop_first, conv=conv, bn=bn, train=training)
    return {"f32": of, "bf16": ob, "pre": None, "prenorm": prenorm}, saved


def transformer_block_fwd(x: torch.Tensor, blk, training: bool, need_dgrad: bool = False,
                          save: Optional[bool] = None, pool_out: Optional[torch.Tensor] = None,
                          prenorm=None, next_norm=None, mask=None, next_blk=None):
    """x fp32 (B, L, d) -> fp32 (B, L, d); returns (out, saved).  ``training``
    switches dropout on; ``save`` (default = training) keeps what backward needs.
    ``pool_out`` (zeroed fp32 (B, d)): the second FFN Linear also accumulates the mean over time of
    the block's output there (the encoder's pooling step, fused into that GEMM's epilogue).
    Width 128: every LayerNorm but the stack's first is computed in the epilogue of the GEMM that produces
    its input (``prenorm`` = (norm1(x) bf16, stats) handed in by the block below, ``next_norm`` = the next
    block's norm1, whose output is then returned in ``saved['next_prenorm']`` / as third result).  With ``next_blk`` the
    same launch also runs that block's QKV projection on those rows (a second GEMM: the third result then carries the
    packed q | k | v as its third element)."""
    B, L, D = x.shape
    M = B * L
    save = training if save is None else save
    p = blk.dropout.p if training else 0.0
    x2 = x.view(M, D)
    fuse_ln = D == 128 and M % 32 == 0
    h1, st1 = prenorm[:2] if prenorm is not None else layernorm(x2, blk.norm1, save)
    if prenorm is not None and len(prenorm) > 2:            # the block below already projected its fused LayerNorm rows
        qkv = prenorm[2]
    else:
        qkv = linear_rows(h1, blk.self_attn.in_proj_weight, blk.self_attn.in_proj_bias,
                          need_dgrad=need_dgrad)["bf16"]
    pa = float(blk.self_attn.dropout) if training else 0.0      # attention-probability dropout
    sa = _next_seed() if pa > 0 else 0
    o, lse = attention(qkv.view(B, L, 3 * D), blk.nhead, save, pa, sa, mask)
    s1 = _next_seed() if p > 0 else 0
    f1 = None
    s2 = None
    if fuse_ln:
        wf, _, cinp, _ = weights.get(blk.self_attn.out_proj.weight, need_dgrad)
        x1 = _empty((M, D), _F32, x)
        h2 = _empty((M, D), _BF, x)
        st2 = _empty((M, 2), _F32, x) if save else None
        w1, _, c1, _ = weights.get(blk.linear1.weight, need_dgrad)
        n1 = blk.linear1.weight.shape[0]
        if c1 == 128 and n1 % 128 == 0 and M * n1 < (1 << 32) and not _NO_FFN1_FUSE:
            # the first FFN Linear (GELU, Dropout, pre-activation copy) runs on norm2's rows inside the out-projection's launch
            s2 = _next_seed() if p > 0 else 0
            g1 = _empty((M, n1), _BF, x)
            z1 = _empty((M, n1), _BF, x) if save else None
            _hip.call("mm_linear_fwd_ln_gemm2_act", o.view(M, D), wf, M, cinp, blk.self_attn.out_proj.bias, x2, x1, float(p),
                      int(s1), EP(), blk.norm2.weight, blk.norm2.bias, float(blk.norm2.eps), h2, st2, w1, blk.linear1.bias,
                      n1, g1, z1, ACT[blk._act], float(p), int(s2))
            f1 = {"bf16": g1, "pre": z1, "f32": None}
        else:
            _hip.call("mm_linear_fwd_ln", o.view(M, D), wf, M, cinp, blk.self_attn.out_proj.bias, x2, x1, float(p), int(s1),
                      EP(), blk.norm2.weight, blk.norm2.bias, float(blk.norm2.eps), h2, st2)
    else:
        x1 = linear_rows(o.view(M, D), blk.self_attn.out_proj.weight, blk.self_attn.out_proj.bias,
                         residual=x2, out_f32=True, out_bf16=False, drop_p=p, seed=s1,
                         need_dgrad=need_dgrad)["f32"]
        h2, st2 = layernorm(x1, blk.norm2, save)
    if f1 is None:
        s2 = _next_seed() if p > 0 else 0
        f1 = linear_rows(h2, blk.linear1.weight, blk.linear1.bias, act=blk._act, out_pre=save,
                         drop_p=p, seed=s2, need_dgrad=need_dgrad)
    s3 = _next_seed() if p > 0 else 0
    nxt = None
    if pool_out is not None:
        wf, _, cinp, _ = weights.get(blk.linear2.weight, need_dgrad)
        x2o = _empty((M, D), _F32, x)
        _hip.call("mm_linear_fwd_meanpool", f1["bf16"], wf, M, cinp, blk.linear2.bias, x1, x2o, float(p), int(s3),
                  EP(), pool_out, L)
    elif next_norm is not None and fuse_ln:
        wf, _, cinp, _ = weights.get(blk.linear2.weight, need_dgrad)
        x2o = _empty((M, D), _F32, x)
        hn = _empty((M, D), _BF, x)
        stn = _empty((M, 2), _F32, x) if save else None
        wq = None
        if next_blk is not None and not _NO_QKV_FUSE:
            wq, _, cq, _ = weights.get(next_blk.self_attn.in_proj_weight, need_dgrad)
            if cq != 128 or next_blk.self_attn.in_proj_weight.shape[0] % 128:
                wq = None
        if wq is not None:
            nq = next_blk.self_attn.in_proj_weight.shape[0]
            qn = _empty((M, nq), _BF, x)
            _hip.call("mm_linear_fwd_ln_gemm2", f1["bf16"], wf, M, cinp, blk.linear2.bias, x1, x2o, float(p), int(s3), EP(),
                      next_norm.weight, next_norm.bias, float(next_norm.eps), hn, stn, wq, next_blk.self_attn.in_proj_bias,
                      nq, qn)
            nxt = (hn, stn, qn)
        else:
            _hip.call("mm_linear_fwd_ln", f1["bf16"], wf, M, cinp, blk.linear2.bias, x1, x2o, float(p), int(s3), EP(),
                      next_norm.weight, next_norm.bias, float(next_norm.eps), hn, stn)
            nxt = (hn, stn)
    else:
        x2o = linear_rows(f1["bf16"], blk.linear2.weight, blk.linear2.bias, residual=x1, out_f32=True,
                          out_bf16=False, drop_p=p, seed=s3, need_dgrad=need_dgrad)["f32"]
    saved = None
    if save:
        saved = dict(x=x2, h1=h1, st1=st1, qkv=qkv, o=o, lse=lse, x1=x1, h2=h2, st2=st2,
                     z=f1["pre"], g=f1["bf16"], p=p, seeds=(s1, s2, s3), attn_drop=(pa, sa), B=B, L=L, blk=blk,
                     mask=mask)
    if next_norm is not None:
        return x2o.view(B, L, D), saved, nxt
    return x2o.view(B, L, D), saved


def pooled_head_fwd(x: torch.Tensor, lin, *, act="gelu", training=False, drop_p=0.0,
                    need_dgrad=False, save=None, pooled_f32=None):
    """mean over L of fp32 (B, L, d) -> Linear -> act [-> dropout]: fp32 (B, out).
    ``pooled_f32``: the mean, when the producer of ``x`` already accumulated it (transformer_block_fwd): a
    (B, 2 d) buffer holding the 64-bit fixed-point accumulator the GEMM epilogue added into."""
    B, L, D = x.shape
    pooled_acc = pooled_f32 is not None
    save = training if save is None else save
    p = drop_p if training else 0.0
    seed = _next_seed() if p > 0 else 0
    if D % 16 == 0 and D <= 1024:                    # fp32 head kernel: one small launch instead of a 1-workgroup GEMM
        if pooled_f32 is None:
            pooled_f32 = _empty((B, D), _F32, x)
            _hip.call("mm_meanpool_fwd", x, pooled_f32, None, B, L, D)
        N = lin.weight.shape[0]
        out = _empty((B, N), _F32, x)
        z = _empty((B, N), _BF, x) if save else None
        pooled = _empty((B, D), _BF, x) if save else None
        _hip.call("mm_pooled_head_fwd", None if pooled_acc else pooled_f32, pooled_f32 if pooled_acc else None,
                  lin.weight, lin.bias, out, z, pooled, B, D, N, ACT[act], float(p), int(seed), EP())
        saved = dict(pooled=pooled, z=z, seed=seed, drop_p=p, B=B, L=L, D=D, lin=lin, act=act, fused=True) if save else None
        return out, saved
    pooled = _empty((B, D), _BF, x)
    _hip.call("mm_meanpool_fwd", x, None, pooled, B, L, D)
    r = linear_rows(pooled, lin.weight, lin.bias, act=act, out_f32=True, out_bf16=False,
                    out_pre=save, drop_p=p, seed=seed, need_dgrad=need_dgrad)
    saved = dict(pooled=pooled, z=r["pre"], seed=seed, drop_p=p, B=B, L=L, D=D, lin=lin,
                 act=act) if save else None
    return r["f32"], saved


def pe_table(pos_encoder, L: int) -> torch.Tensor:
    """(L, d) fp32 view of the sinusoid buffer (max_len, 1, d)."""
    pe = pos_encoder.pe
    if L > pe.shape[0]:
        raise ValueError(f"sequence length {L} exceeds PositionalEncoding max_len {pe.shape[0]}")
    return pe[:L, 0, :].contiguous()


# ------------------------------------------------------------ EEG encoders
def _encoder_tail_impl(m, h, training: bool, need_dgrad: bool, save: bool, prenorm=None, stages=None):
    """shared tail of both EEG encoders: transformer stack -> mean pool -> Linear -> GELU.
    ``prenorm``: (norm1(h) bf16, stats) of the FIRST block when the producer of ``h`` already formed it.
    ``stages`` (dict, inspection only): receives a copy of the residual stream after every block (``block<i>``)."""
    blocks = []
    B, L, D = h.shape
    nblk = len(m.transformer_layers)
    # 64-bit fixed-point mean accumulator (one replica): the last block's GEMM epilogue adds into it
    pooled = _zeros((B, 2 * D), h) if nblk and L % 32 == 0 and D == 128 else None
    layers = list(m.transformer_layers)
    for i, blk in enumerate(layers):
        if i < nblk - 1:                                 # the next block's norm1 rides in this block's last GEMM
            h, s, prenorm = transformer_block_fwd(h, blk, training, need_dgrad, save=save, prenorm=prenorm,
                                                  next_norm=layers[i + 1].norm1, next_blk=layers[i + 1])
        else:
            h, s = transformer_block_fwd(h, blk, training, need_dgrad, save=save, pool_out=pooled, prenorm=prenorm)
        blocks.append(s)
        if stages is not None:
            stages[f"block{i}"] = h.clone()
    out, s = pooled_head_fwd(h, m.output_proj[2], training=training, drop_p=m.drop_p, need_dgrad=need_dgrad,
                             save=save, pooled_f32=pooled)
    return out, blocks, s, h


def _erp_forward_impl(m, x: torch.Tensor, training: bool, need_dgrad: bool, save=None, xb=None, stages=None):
    """EnhancedERPEncoder forward; returns (features fp32 (B, H), saved list).
    ``save`` (default = training): keep what a backward needs (eval + save = frozen BatchNorm).
    ``xb``: ``pack_nct(x)`` when the caller already holds it (a trainer stages its inputs packed).
    ``stages`` (dict, inspection only - the parity tests hold the HIP path's intermediate activations against the
    reference's, enhanced_models_v4.py:169-193): receives channels-last copies of what the path materialises - ``conv1``
    / ``conv2`` (bf16 (B, T, C)), ``pos`` (fp32, conv block 3 + positional table, one launch) and ``block<i>`` - plus
    ``conv3``, which the fused launch never forms: one EXTRA launch of the same kernel without the table."""
    cl = m.conv_layers
    save = training if save is None else save
    p = m.drop_p if training else 0.0
    if xb is None:
        xb = pack_nct(x)
    saved = []
    saved_in = []                                        # conv block outputs (for ``stages``)
    r, s = conv_bn_act(xb, cl[0], cl[1], training=training, drop_p=p, need_dgrad=need_dgrad, save=save)
    saved.append(s)
    saved_in.append(r["bf16"])
    r, s = conv_bn_act(r["bf16"], cl[4], cl[5], pool=2, training=training, drop_p=p,
                       drop_first=False, need_dgrad=need_dgrad, save=save)
    saved.append(s)
    saved_in.append(r["bf16"])
    L = r["bf16"].shape[1]
    r, s = conv_bn_act(r["bf16"], cl[9], cl[10], training=training, drop_p=p,
                       pe=pe_table(m.pos_encoder, L), pe_drop_p=(m.pos_encoder.dropout.p if training else 0.0),
                       want_f32=True, want_bf16=False, need_dgrad=need_dgrad, save=save,
                       next_norm=m.transformer_layers[0].norm1 if len(m.transformer_layers) else None)
    saved.append(s)
    if stages is not None:
        stages["conv1"], stages["conv2"] = saved_in[0].clone(), saved_in[1].clone()
        stages["pos"] = r["f32"].clone()
        if not save:                                     # (the fused eval launch: BatchNorm folded into the GEMM epilogue)
            wf3 = weights.get(cl[9].weight, need_dgrad)[0]
            o4 = bn_fold_eval(cl[10], cl[9].bias)
            stages["conv3"] = igemm(saved_in[1], wf3, cl[9].kernel_size[0], cl[9].padding[0], cl[9].out_channels,
                                    scale=o4[0], shift=o4[1], act="gelu", out_f32=True, out_bf16=False)["f32"]
    out, blocks, s, h = _encoder_tail_impl(m, r["f32"], training, need_dgrad, save, prenorm=r.get("prenorm"), stages=stages)
    return out, dict(convs=saved, blocks=blocks, head=s, x_shape=tuple(x.shape), tokens=h)


def erp_encoder_forward(m, x: torch.Tensor) -> torch.Tensor:
    _need_gpu(x)
    if m.training or _wants_grad(m, x):
        from .autograd import ErpEncoderFn
        return ErpEncoderFn.run(m, x)
    with torch.no_grad():
        return _erp_forward_impl(m, x.float(), False, False)[0]


def _wants_grad(m, x) -> bool:
    """a backward may follow: autograd is on and the input or a parameter asks for a gradient"""
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in m.parameters()))


def add_positional(x, pe, drop_p, training):
    """PositionalEncoding.forward on the fp32 stream (reference quirk kept, enhanced_models_v4.py:49:
    a 3-D input is batch-first unless ``x.size(1) == 1``, which is read as (seq, batch=1, d))."""
    _need_gpu(x)
    if x.dim() != 3:
        raise ValueError("PositionalEncoding: expected a 3-D input (batch, seq, d_model) or (seq, 1, d_model)")
    L = x.size(1) if x.size(1) != 1 else x.size(0)
    if L > pe.shape[0]:
        raise ValueError(f"sequence length {L} exceeds PositionalEncoding max_len {pe.shape[0]}")
    from .autograd import AddPositionalFn
    return AddPositionalFn.apply(x, pe, float(drop_p) if training else 0.0)


def add_positional_impl(x, pe, p: float, seed: int, backward: bool = False):
    """one mm_add_pe launch; x (B, L, D) batch-first or (L, 1, D); ``backward``: mask only."""
    shape = x.shape
    if x.size(1) != 1:
        B, L, D = shape
    else:
        L, B, D = shape[0], 1, shape[2]
    xf = x.float().contiguous()
    tab = None if backward else pe[:L, 0, :].contiguous()
    out = _empty(tuple(shape), _F32, xf)
    _hip.call("mm_add_pe", xf, tab, out, None, B, L, D, float(p), int(seed), EP())
    return out


def transformer_block(x, blk, training, mask=None):
    _need_gpu(x)
    mask = additive_attn_mask(mask, x.shape[1], x)
    if training or (torch.is_grad_enabled() and x.requires_grad):
        from .autograd import TransformerBlockFn
        return TransformerBlockFn.run(blk, x, mask)
    with torch.no_grad():
        return transformer_block_fwd(x.float().contiguous(), blk, False, mask=mask)[0]


# --------------------------------------------------------- 3-D voxel encoder
def pack_volume(x: torch.Tensor) -> torch.Tensor:
    """(B, 1, D, H, W) fp32 -> (B, D, H, W, 16) bf16 (channel 0 = voxel)."""
    B, C, D, H, W = x.shape
    if C != 1:
        raise _hip.HipLibraryError("fMRIVolumeEncoder3D: single-channel volumes only")
    y = _empty((B, D, H, W, 16), _BF, x)
    _hip.call("mm_pack_volume_bf16", x.contiguous(), y, B * D * H * W, 16)
    return y


def conv3d_bn_act(xv: torch.Tensor, conv, bn, *, pool: bool, training: bool, drop_p: float,
                  need_dgrad: bool, save=None):
    """Conv3d(k3,p1) -> BatchNorm3d -> GELU [-> MaxPool3d(2)] [-> Dropout] on
    channels-last bf16 volumes.  Returns (out, saved).  A pooled layer keeps its pre-BatchNorm
    tensor in bf16 (half the traffic of the conv's output, the pooling pass and the backward; the
    statistics come from the fp32 accumulators); the un-pooled last layer keeps fp32.
    eval with ``save`` (a backward will follow: frozen BatchNorm - fine-tuning on a frozen encoder, saliency):
    the train-shaped pipeline with the RUNNING statistics, no statistic update, no dropout."""
    save = training if save is None else save
    B, D, H, W, cinp = xv.shape
    cout = conv.out_channels
    wf, _, cp, _ = weights.get(conv.weight, need_dgrad)
    assert cp == cinp, (cp, cinp)
    y = _empty((B, D, H, W, cout), _BF if pool else _F32, xv)
    yf, yb = (None, y) if pool else (y, None)
    fin = None
    if training:
        stats = _zeros((REPL, 2, cout), xv)
        cal = kernel_timer.bracket(f"event_pair_c{cinp}")    # an EMPTY bracket on the same stream: what a pair of
        if cal is not None:                                  # HIP event records costs by itself (calibration)
            cal.record()
        end = kernel_timer.bracket(f"conv3d_fwd_c{cinp}")
        _hip.call("mm_conv3d_fwd", xv, wf, B, D, H, W, cinp, cout, conv.bias, stats, yf, yb)
        if end is not None:
            end.record()
        if cout <= 256 and not _NO_FIN_FOLD:           # the finalize runs in the prologue of the apply pass below
            fin, out4 = bn_fin_desc(bn, stats, B * D * H * W)
        else:
            out4 = bn_finalize_train(bn, stats, B * D * H * W)
    elif save:
        _hip.call("mm_conv3d_fwd", xv, wf, B, D, H, W, cinp, cout, conv.bias, None, yf, yb)
        out4 = bn_fold_eval(bn, None)
    else:
        _hip.call("mm_conv3d_fwd", xv, wf, B, D, H, W, cinp, cout, None, None, yf, yb)
        out4 = bn_fold_eval(bn, conv.bias)
    seed = _next_seed() if (training and drop_p > 0) else 0
    p = drop_p if training else 0.0
    ysel = arg = None
    fold = training and fin is not None
    if pool:
        out = _empty((B, D // 2, H // 2, W // 2, cout), _BF, xv)
        if save:                                       # the window winners: all that BN-backward's reduction needs
            ysel = _empty(out.shape, _BF, xv)
            arg = _empty(out.shape, torch.uint8, xv)
        if fold:
            import ctypes
            _hip.call("mm_pool3d_bn_act_fwd_fin", y, ctypes.addressof(fin), out, ysel, arg, B, D, H, W, cout, ACT["gelu"], float(p), seed, EP())
        else:
            _hip.call("mm_pool3d_bn_act_fwd", y, out4, out, ysel, arg, B, D, H, W, cout, ACT["gelu"], float(p), seed, EP())
    else:
        out = _empty((B, D * H * W, cout), _F32, xv)
        if fold:
            import ctypes
            _hip.call("mm_bn_act_fwd_fin", y, ctypes.addressof(fin), None, None, out, B, D * H * W, cout,
                      ACT["gelu"], 1, 1, float(p), seed, 0.0, 0, EP())
        else:
            _hip.call("mm_bn_act_fwd", y, out4[0], out4[1], None, None, out, B, D * H * W, cout,
                      ACT["gelu"], 1, 1, float(p), seed, 0.0, 0, EP())
    saved = dict(xv=xv, y=y, out4=out4, pool=pool, drop_p=p, seed=seed, conv=conv, bn=bn,
                 ysel=ysel, arg=arg, train=training) if save else None
    return out, saved


def conv3d_l1_bn_act(x: torch.Tensor, conv, bn, *, training: bool, drop_p: float, save=None, winners=None):
    """fused first layer on the raw fp32 volume (B, 1, D, H, W); see conv3d_l1.hip.  ``save`` without ``training``:
    frozen BatchNorm (running statistics), a backward will follow.  ``winners`` (list, inspection only): receives the
    u8 (B, D/2, H/2, W/2, 32) tensor of pooling-window winners the forward launch then also writes."""
    save = training if save is None else save
    B, _, D, H, W = x.shape
    wimg = weights.get(conv.weight.view(32, 27, 1), False, key=conv.weight)[0]
    out = _empty((B, D // 2, H // 2, W // 2, 32), _BF, x)
    p = drop_p if training else 0.0
    seed = _next_seed() if p > 0 else 0
    gramc = fin = None
    if training:
        # BatchNorm statistics from the Gram matrix of the im2col matrix (no per-channel pass over the convolution):
        # csrc/conv3d_l1.hip.  The workspace is kept: the backward's weight-gradient correction terms come from it.
        gram = _zeros((REPL, 32, 32), x)
        stats = _zeros((REPL, 2, 32), x)
        _hip.call("mm_conv3d_l1_gram", x, wimg, conv.bias, gram, stats, B, D, H, W)
        gramc = gram                                   # (the accumulator workspace itself: the backward's combine step reads it)
        if not _NO_FIN_FOLD and winners is None:       # the finalize runs in the forward kernel's prologue
            fin, out4 = bn_fin_desc(bn, stats, B * D * H * W)
        else:
            out4 = bn_finalize_train(bn, stats, B * D * H * W)
        bias = conv.bias
    elif save:
        out4 = bn_fold_eval(bn, None)                  # the backward recomputes conv + bias and needs mean / rstd apart
        bias = conv.bias
    else:
        out4 = bn_fold_eval(bn, conv.bias)
        bias = None
    if training and fin is not None:
        import ctypes
        _hip.call("mm_conv3d_l1_fwd_fin", x, wimg, bias, ctypes.addressof(fin), out, B, D, H, W, float(p), seed, EP())
    elif winners is not None:
        arg = _empty(tuple(out.shape), torch.uint8, x)
        _hip.call("mm_conv3d_l1_fwd_winners", x, wimg, bias, out4, out, arg, B, D, H, W, 1 if training else 0, float(p), seed, EP())
        winners.append(arg)
    else:
        _hip.call("mm_conv3d_l1", 1, x, wimg, bias, out4, None, None, None, out, None, None,
                  B, D, H, W, 1 if training else 0, float(p), seed, EP())
    saved = dict(l1=True, x=x, wimg=wimg, out4=out4, drop_p=p, seed=seed, conv=conv, bn=bn, train=training, gramc=gramc) if save else None
    return out, saved


def _vol_forward_impl(m, x: torch.Tensor, training: bool, need_dgrad: bool, save=None, need_dx: bool = False, winners=None):
    """``save`` (default = training): keep what a backward needs; eval + save = frozen BatchNorm.  ``need_dx``: the
    caller wants d / d volume - layer 1 then runs as an ordinary implicit GEMM on the channel-padded volume (the
    fused layer-1 kernels never form its data gradient).  ``winners`` (list, inspection only; needs ``save``): receives
    the two max-pool layers' window winners, u8 channels-last, j = (dd << 2) | (hh << 1) | ww."""
    save = training if save is None else save
    cl = m.conv_layers
    p = m.drop_p
    saved = []
    B, C, D, H, W = x.shape
    x = x.contiguous()
    if C == 1 and cl[0].out_channels == 32 and D % 2 == 0 and H % 2 == 0 and W % 2 == 0 and not need_dx:
        h, s = conv3d_l1_bn_act(x, cl[0], cl[1], training=training, drop_p=p, save=save, winners=winners)
    else:
        h, s = conv3d_bn_act(pack_volume(x), cl[0], cl[1], pool=True, training=training, drop_p=p,
                             need_dgrad=need_dgrad or need_dx, save=save)
    saved.append(s)
    h, s = conv3d_bn_act(h, cl[5], cl[6], pool=True, training=training, drop_p=p, need_dgrad=need_dgrad, save=save)
    saved.append(s)
    if winners is not None:
        if len(winners) == 0:                            # layer 1 took the generic path: its winners are saved like layer 2's
            winners.append(saved[0]["arg"])
        winners.append(s["arg"])
    h, s = conv3d_bn_act(h, cl[10], cl[11], pool=False, training=training, drop_p=p, need_dgrad=need_dgrad, save=save)
    saved.append(s)
    out, hs = pooled_head_fwd(h, m.output_proj[2], training=training, drop_p=p, need_dgrad=need_dgrad, save=save)
    return out, dict(convs=saved, head=hs, x_shape=tuple(x.shape), need_dx=need_dx)


def volume_encoder_forward(m, x: torch.Tensor) -> torch.Tensor:
    _need_gpu(x)
    if m.training or _wants_grad(m, x):
        from .autograd import VolumeEncoderFn
        return VolumeEncoderFn.run(m, x)
    with torch.no_grad():
        return _vol_forward_impl(m, x.float(), False, False)[0]


# ------------------------------------------------- projection bridge (fp32)
def small_linear(x: torch.Tensor, lin, *, act="none", drop_p=0.0, seed=0, want_pre=False, bn=None,
                 weight=None, bias=None):
    """fp32 (B, K) -> (B, N): dropout(act((x W^T + b) * scale + shift)); scale/shift
    = eval-mode BatchNorm1d ``bn`` folded; optional pre-activation copy."""
    W = lin.weight if weight is None else weight
    bvec = (lin.bias if lin is not None else None) if bias is None else bias
    B, K = x.shape
    N = W.shape[0]
    y = _empty((B, N), _F32, x)
    pre = _empty((B, N), _F32, x) if want_pre else None
    sc = sh = None
    if bn is not None:
        out4 = bn_fold_eval(bn, None)
        sc, sh = out4[0], out4[1]
    _hip.call("mm_small_linear_fwd", x, W, bvec, sc, sh, y, pre, B, K, N, ACT[act], float(drop_p), int(seed), EP())
    return y, pre


def proj_head_fwd(seq, x: torch.Tensor, training: bool, drop_p: float):
    """Linear -> LayerNorm -> GELU -> Dropout (bridge_utils.py:34-45) in fp32."""
    lin, ln = seq[0], seq[1]
    B = x.shape[0]
    N = lin.weight.shape[0]
    z1, _ = small_linear(x, lin)
    hn = _empty((B, N), _F32, x)
    stat = _empty((B, 2), _F32, x)
    _hip.call("mm_layernorm_fwd", z1, ln.weight, ln.bias, None, hn, stat, B, N, float(ln.eps))
    p = drop_p if training else 0.0
    seed = _next_seed() if p > 0 else 0
    a = _empty((B, N), _F32, x)
    _hip.call("mm_act_f32", hn, a, B * N, ACT["gelu"], float(p), seed, EP())
    return a, dict(x=x, z1=z1, hn=hn, stat=stat, p=p, seed=seed, seq=seq)


def contrastive_embed_impl(bridge, eeg: torch.Tensor, fmri: torch.Tensor, training: bool):
    """-> packed L2-normalised embeddings z (B, 2N) = [ze | zf], saved."""
    B = eeg.shape[0]
    N = bridge.bridge_dim
    xe, xf = eeg.float().contiguous(), fmri.float().contiguous()
    (le, lne), (lf, lnf) = bridge.eeg_proj[:2], bridge.fmri_proj[:2]
    assert lne.eps == lnf.eps
    p = float(bridge.drop_p) if training else 0.0
    se, sf = (_next_seed(), _next_seed()) if p > 0 else (0, 0)
    z = _empty((B, 2 * N), _F32, eeg)
    nrm = _empty((2, B), _F32, eeg)
    z1 = _empty((2, B, N), _F32, eeg)
    hn = _empty((2, B, N), _F32, eeg)
    stat = _empty((2, B, 2), _F32, eeg)
    _hip.call("mm_proj_heads_fwd", xe, le.weight, le.bias, lne.weight, lne.bias, xe.shape[1],
              xf, lf.weight, lf.bias, lnf.weight, lnf.bias, xf.shape[1],
              z1, hn, stat, z, nrm, B, N, float(lne.eps), p, se, sf, EP())
    return z, dict(xe=xe, xf=xf, z1=z1, hn=hn, stat=stat, z=z, nrm=nrm, B=B, N=N, p=p, seeds=(se, sf), bridge=bridge)


def contrastive_embed(bridge, eeg, fmri, training):
    _need_gpu(eeg, fmri)
    from .autograd import ContrastiveEmbedFn
    z = ContrastiveEmbedFn.run(bridge, eeg, fmri)
    N = bridge.bridge_dim
    return z[:, :N], z[:, N:]


def clip_loss(ze, zf, logit_scale, group=None):
    """symmetric InfoNCE over the (all-gathered) batch -> (loss, top1 e->f, top1 f->e)."""
    _need_gpu(ze)
    from .autograd import ClipLossFn
    return ClipLossFn.apply(_packed_pair(ze, zf), logit_scale, group)


def _packed_pair(ze: torch.Tensor, zf: torch.Tensor) -> torch.Tensor:
    """(B, 2N) [ze | zf]: the shared base when ze / zf are EXACTLY its two column halves (what
    contrastive_embed returns), a concatenation for anything else (row subsets, other views)."""
    base = ze._base
    if (base is not None and base is zf._base and base.dim() == 2 and base.is_contiguous()
            and ze.dim() == 2 and ze.shape == zf.shape and base.shape == (ze.shape[0], 2 * ze.shape[1])
            and ze.stride() == base.stride() and zf.stride() == base.stride()
            and ze.storage_offset() == base.storage_offset()
            and zf.storage_offset() == base.storage_offset() + ze.shape[1]):
        return base
    return torch.cat([ze, zf], dim=1)



# ----------------------------------------------------- small models (forward)
def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


def learned_fusion(m, feats: List[torch.Tensor], training: bool, autograd: bool = False):
    """LearnedFusionModule.forward -> (fused (B, H), weights (B, M)).  ``autograd``: differentiable
    composition even in eval mode (its gate dropout is then off)."""
    _need_gpu(*feats)
    if training or autograd:
        from . import small_autograd as sa
        g = sa.linear(torch.cat(list(feats), dim=1), m.gate_net[0], "gelu", m.gate_net[2].p if training else 0.0)
        dyn = sa.linear(g, m.gate_net[3])
        return sa.LearnedFusionFn.apply(dyn, m.fusion_logits, m.temperature, *feats)
    M = len(feats)
    fs = [_f32c(f) for f in feats]
    B, H = fs[0].shape
    g, _ = small_linear(torch.cat(fs, dim=1), m.gate_net[0], act="gelu")
    dyn, _ = small_linear(g, m.gate_net[3])
    fused = _empty((B, H), _F32, fs[0])
    w = _empty((B, M), _F32, fs[0])
    _hip.call("mm_learned_fusion", fs[0], fs[1], fs[2] if M > 2 else None, dyn, _f32c(m.fusion_logits),
              _f32c(m.temperature).reshape(1), fused, w, B, H, M)
    return fused, w


def bridge_forward(m, eeg, fmri):
    """EEGfMRIBridgeFusionNet.forward -> (logits, fused, fusion_w (B,2), attn_w (B,1,2))."""
    _need_gpu(eeg, fmri)
    # eval mode with a backward to follow (gradient saliency / integrated gradients,
    # bridge_utils.py:158-229): the autograd composition with every dropout off - the bridge has
    # LayerNorms only, so that IS the eval forward
    wants = torch.is_grad_enabled() and (eeg.requires_grad or fmri.requires_grad)
    if m.training or wants:
        from . import small_autograd as sa
        p = m.drop_p if m.training else 0.0
        ca = m.cross_attn
        ep = sa.proj_head(eeg, m.eeg_proj, p)
        fp = sa.proj_head(fmri, m.fmri_proj, p)
        pe = sa.SmallLinearFn.apply(ep, ca.in_proj_weight, ca.in_proj_bias, "none", 0.0)
        pf = sa.SmallLinearFn.apply(fp, ca.in_proj_weight, ca.in_proj_bias, "none", 0.0)
        ctx, attw = sa.Attn1x2Fn.apply(pe, pf, m.num_heads, float(ca.dropout) if m.training else 0.0)
        att = sa.linear(ctx, ca.out_proj)
        fused, fw = learned_fusion(m.fusion, [att, fp], m.training, autograd=True)
        c = m.classifier
        h = sa.ActFn.apply(sa.LayerNormFn.apply(sa.linear(fused, c[0]), c[1].weight, c[1].bias, c[1].eps), "relu", p)
        return sa.linear(h, c[4]), fused, fw, attw.view(-1, 1, 2)
    with torch.no_grad():
        ep, _ = proj_head_fwd(m.eeg_proj, _f32c(eeg), False, 0.0)
        fp, _ = proj_head_fwd(m.fmri_proj, _f32c(fmri), False, 0.0)
        B, E = ep.shape
        ca = m.cross_attn
        pe, _ = small_linear(ep, None, weight=ca.in_proj_weight, bias=ca.in_proj_bias)
        pf, _ = small_linear(fp, None, weight=ca.in_proj_weight, bias=ca.in_proj_bias)
        ctx = _empty((B, E), _F32, ep)
        attw = _empty((B, 2), _F32, ep)
        _hip.call("mm_attn_1x2", pe, pf, ctx, attw, B, E, m.num_heads)
        att, _ = small_linear(ctx, ca.out_proj)
        fused, fw = learned_fusion(m.fusion, [att, fp], False)
        h1, _ = small_linear(fused, m.classifier[0])
        ln = m.classifier[1]
        hn = _empty(h1.shape, _F32, h1)
        _hip.call("mm_layernorm_fwd", h1, ln.weight, ln.bias, None, hn, None, B, h1.shape[1], float(ln.eps))
        hr = _empty(h1.shape, _F32, h1)
        _hip.call("mm_act_f32", hn, hr, hn.numel(), ACT["relu"], 0.0, 0, None)
        logits, _ = small_linear(hr, m.classifier[4])
    return logits, fused, fw, attw.view(B, 1, 2)


def _mlp_bn_act(x, lin, bn, act, drop_p, training):
    """Linear -> BatchNorm1d -> act -> Dropout on fp32 rows (eval: BN folded into the linear)."""
    if training:
        from . import small_autograd as sa
        return sa.linear_bn_act(x, lin, bn, act, drop_p)
    if torch.is_grad_enabled() and (x.requires_grad or lin.weight.requires_grad):
        # eval mode with a backward to follow: frozen BatchNorm (running statistics, no update), dropout off
        from . import small_autograd as sa
        return sa.linear_bn_act(x, lin, bn, act, 0.0, frozen=True)
    return small_linear(_f32c(x), lin, act=act, bn=bn)[0]


def small_autograd_linear(x, lin, act="none", drop_p=0.0):
    """differentiable fp32 Linear (+ activation, dropout) on (B, K) rows: small_autograd.linear"""
    from . import small_autograd as sa
    return sa.linear(x, lin, act, drop_p)


def _v4_classifier(cl, fused, p, training):
    from . import small_autograd as sa
    h = _mlp_bn_act(fused, cl[0], cl[1], "gelu", p, training)
    h = _mlp_bn_act(h, cl[4], cl[5], "gelu", p, training)
    return sa.linear(h, cl[8])


def trimodal_v4_forward(m, erp, pw, conn):
    """EnhancedTriModalFusionNetV4.forward (crossmodal_v4_enhancements.py:338-388)
    -> (logits, fusion weights (B, 3), fused (B, H))"""
    _need_gpu(erp, pw, conn)
    from . import small_autograd as sa
    tr, p = m.training, m.drop_p
    e = m.erp_encoder(erp)
    w = m.pw_encoder(pw)
    c = conn.reshape(conn.size(0), -1) if conn.dim() > 2 else conn
    ce = m.conn_encoder
    c = _mlp_bn_act(c, ce[0], ce[1], "gelu", p, tr)
    c = _mlp_bn_act(c, ce[4], ce[5], "gelu", p, tr)
    enh, _ = sa.mha_1xk(m.cross_attn, [e, w, c], tr)
    fused, weights = learned_fusion(m.fusion, [enh, w, c], tr)
    return _v4_classifier(m.classifier, fused, p, tr), weights, fused


def bidirectional_cross_attention_forward(m, e, w):
    """BiDirectionalCrossAttention.forward (crossmodal_v4_enhancements.py:436-466)"""
    _need_gpu(e, w)
    from . import small_autograd as sa
    tr = m.training
    p = m.dropout.p if tr else 0.0
    ea, _ = sa.mha_1xk(m.erp_to_pw_attn, [e, w], tr)
    wa, _ = sa.mha_1xk(m.pw_to_erp_attn, [w, e], tr)       # softmax over the same two keys: order is immaterial

    def gated(feat, att, gate, norm):
        g = sa.linear(torch.cat([_f32c(feat), att], dim=1), gate[0], "sigmoid")
        upd = sa.ActFn.apply(sa.MulFn.apply(g, att), "none", p)
        return sa.LayerNormFn.apply(sa.AddFn.apply(feat, upd), norm.weight, norm.bias, norm.eps)
    return gated(e, ea, m.erp_gate, m.norm_erp), gated(w, wa, m.pw_gate, m.norm_pw)


def smart_fusion_v4_forward(m, erp, pw):
    """EnhancedSmartFusionNetV4.forward (crossmodal_v4_enhancements.py:537-570)"""
    _need_gpu(erp, pw)
    e = m.erp_encoder(erp)
    w = m.pw_encoder(pw)
    if m.use_cross_attention:
        e, w = m.cross_attention(e, w)
    fused, weights = learned_fusion(m.fusion, [e, w], m.training)
    return _v4_classifier(m.classifier, fused, m.drop_p, m.training), weights, fused


def fmri_mlp_forward(seq, x, drop_p, training):
    """Linear-BN-ReLU-Drop x2 (fmri_utils.py:26-35); eval-mode BN is folded."""
    _need_gpu(x)
    if training:
        from . import small_autograd as sa
        h = sa.linear_bn_act(x, seq[0], seq[1], "relu", drop_p)
        return sa.linear_bn_act(h, seq[4], seq[5], "relu", drop_p)
    h, _ = small_linear(_f32c(x), seq[0], act="relu", bn=seq[1])
    h, _ = small_linear(h, seq[4], act="relu", bn=seq[5])
    return h


def fmri_single_forward(m, x):
    """fMRIActivationOnly / fMRIConnectivityOnly (run_fmri_v11.py:311-370): encoder MLP -> Linear-ReLU-Dropout-Linear."""
    _need_gpu(x)
    if m.training:
        from . import small_autograd as sa
        p = m.drop_p
        feat = fmri_mlp_forward(m.encoder.encoder, x, p, True)
        return sa.linear(sa.linear(feat, m.head[0], "relu", p), m.head[3])
    with torch.no_grad():
        feat = fmri_mlp_forward(m.encoder.encoder, x, 0.0, False)
        h, _ = small_linear(feat, m.head[0], act="relu")
        out, _ = small_linear(h, m.head[3])
    return out


def fmri_fusion_forward(m, activation, connectivity):
    _need_gpu(activation, connectivity)
    if m.training:
        from . import small_autograd as sa
        p = m.drop_p
        a = fmri_mlp_forward(m.activation_encoder.encoder, activation, p, True)
        c = fmri_mlp_forward(m.connectivity_encoder.encoder, connectivity, p, True)
        comb = sa.Softmax2ConcatFn.apply(a, c, m.activation_weight, m.connectivity_weight)
        fused = sa.linear_bn_act(comb, m.fusion[0], m.fusion[1], "relu", p)
        return sa.linear(sa.linear(fused, m.head[0], "relu", p), m.head[3]), fused
    with torch.no_grad():
        a = fmri_mlp_forward(m.activation_encoder.encoder, activation, 0.0, False)
        c = fmri_mlp_forward(m.connectivity_encoder.encoder, connectivity, 0.0, False)
        B, H = a.shape
        comb = _empty((B, 2 * H), _F32, a)
        _hip.call("mm_softmax2_concat", a, c, _f32c(m.activation_weight), _f32c(m.connectivity_weight), comb, B, H, H)
        fused, _ = small_linear(comb, m.fusion[0], act="relu", bn=m.fusion[1])
        h, _ = small_linear(fused, m.head[0], act="relu")
        out, _ = small_linear(h, m.head[3])
    return out, fused


def conn_encoder_forward(m, x):
    _need_gpu(x)
    if m.training:
        from . import small_autograd as sa
        p = m.drop_p
        h = sa.linear_bn_act(x, m.proj1[0], m.proj1[1], "gelu", p)
        h = sa.linear_bn_act(h, m.proj2[0], m.proj2[1], "gelu", p)
        gate = sa.linear(sa.linear(h, m.attention[0], "tanh"), m.attention[2], "sigmoid")
        return sa.linear_bn_act(sa.MulFn.apply(h, gate), m.output[0], m.output[1], "gelu", p)
    with torch.no_grad():
        h, _ = small_linear(_f32c(x), m.proj1[0], act="gelu", bn=m.proj1[1])
        h, _ = small_linear(h, m.proj2[0], act="gelu", bn=m.proj2[1])
        t, _ = small_linear(h, m.attention[0], act="tanh")
        g, _ = small_linear(t, m.attention[2], act="sigmoid")
        hg = torch.empty_like(h)
        _hip.call("mm_mul_f32", h, g, hg, h.numel())
        out, _ = small_linear(hg, m.output[0], act="gelu", bn=m.output[1])
    return out


def hybrid_fusion_forward(m, erp, pw, conn):
    _need_gpu(erp, pw, conn)
    if m.training:
        from . import small_autograd as sa
        p = m.drop_p
        g = sa.linear(sa.linear(torch.cat([erp, pw], dim=1), m.erp_pw_gate[0], "gelu", p), m.erp_pw_gate[3])
        comb, gate = sa.Gate2MixFn.apply(g, erp, pw, conn, float(m.conn_boost))
        return sa.linear_bn_act(comb, m.late_fusion[0], m.late_fusion[1], "gelu", p), gate
    with torch.no_grad():
        e, p, c = _f32c(erp), _f32c(pw), _f32c(conn)
        B, H = e.shape
        g1, _ = small_linear(torch.cat([e, p], dim=1), m.erp_pw_gate[0], act="gelu")
        g2, _ = small_linear(g1, m.erp_pw_gate[3])
        comb = _empty((B, 2 * H), _F32, e)
        gate = _empty((B, 2), _F32, e)
        _hip.call("mm_gate2_mix", g2, e, p, c, comb, gate, B, H, float(m.conn_boost))
        fused, _ = small_linear(comb, m.late_fusion[0], act="gelu", bn=m.late_fusion[1])
    return fused, gate


def bn_classifier_forward(seq, fused, drop_p, training):
    """Linear-BN-GELU-Drop-Linear (crossmodal_v4_enhancements.py:909-915)."""
    _need_gpu(fused)
    if training:
        from . import small_autograd as sa
        return sa.linear(sa.linear_bn_act(fused, seq[0], seq[1], "gelu", drop_p), seq[4])
    if torch.is_grad_enabled() and (fused.requires_grad or seq[0].weight.requires_grad):
        from . import small_autograd as sa                   # eval + backward: frozen BatchNorm, dropout off
        return sa.linear(sa.linear_bn_act(fused, seq[0], seq[1], "gelu", 0.0, frozen=True), seq[4])
    with torch.no_grad():
        h, _ = small_linear(_f32c(fused), seq[0], act="gelu", bn=seq[1])
        out, _ = small_linear(h, seq[4])
    return out


def lite_encoder_forward(m, x):
    """LiteERPEncoder / LitePowerEncoder (crossmodal_v4_enhancements.py:817-877)."""
    _need_gpu(x)
    if m.training:
        from . import small_autograd as sa
        pooled = sa.LiteConvFn.apply(m, x, *list(m.conv_layers.parameters()))
        return sa.linear(pooled, m.output[1], "gelu", m.drop_p)
    with torch.no_grad():
        cl = m.conv_layers
        xb = pack_nct(x.float())
        r, _ = conv_bn_act(xb, cl[0], cl[1], pool=2, training=False)
        r, _ = conv_bn_act(r["bf16"], cl[5], cl[6], training=False)
        h = r["bf16"]
        B, T, N = h.shape
        pooled = _empty((B, N), _F32, h)
        _hip.call("mm_meanpool_bf16", h, pooled, B, T, N)
        out, _ = small_linear(pooled, m.output[1], act="gelu")
    return out


def _power_merged(m, device):
    """the three parallel Conv1d(C->64, k in {3,5,7}) as ONE k=7 conv with 192
    outputs (shorter kernels zero-padded around the centre tap) + folded BN."""
    ws, bs, o4 = [], [], []
    for seq in (m.conv_scale1, m.conv_scale2, m.conv_scale3):
        w = seq[0].weight.detach()
        k = w.shape[2]
        ws.append(torch.nn.functional.pad(w, ((7 - k) // 2, (7 - k) // 2)))
        o4.append(bn_fold_eval(seq[1], seq[0].bias))
    return torch.cat(ws, dim=0).contiguous(), torch.cat(o4, dim=1).contiguous()


def stft_front_end(x: torch.Tensor, n_ffts, hop: int, normalize: bool = False, keep_spec: bool = False):
    """(B, C, T) fp32 -> channels-last bf16 (B, frames, Cp) multi-scale STFT power,
    channel order [scale][c][f] as torch.cat([stft_power(x, n) ...], dim=1).
    ``normalize``: z-score every sample's spectra (the reference's normalize_modality on its power
    features, run_training_lite.py:48-51, 162) in fp32 before the bf16 cast."""
    B, C, T = x.shape
    widths = [C * (n // 2 + 1) for n in n_ffts]
    total = sum(widths)
    cp = cpad(total)
    frames = T // hop + 1
    xc = x.float().contiguous()
    if normalize:
        spec = torch.zeros((B, frames, cp), dtype=_F32, device=x.device) if cp != total else _empty((B, frames, cp), _F32, x)
        out = _empty((B, frames, cp), _BF, x)
    else:
        spec = None
        out = torch.zeros((B, frames, cp), dtype=_BF, device=x.device) if cp != total else _empty((B, frames, cp), _BF, x)
    off = 0
    for n, wdt in zip(n_ffts, widths):
        _hip.call("mm_stft_power", xc, None if normalize else out, spec, B, C, T, int(n), int(hop), off, cp)
        off += wdt
    if normalize:
        _hip.call("mm_sample_zscore_bf16", spec, out, _empty((64 * B,), torch.float64, x), B, frames, total, cp, 1e-8)
    return (out, spec) if keep_spec else out


def stft_power_encoder_forward(m, x):
    """MultiScaleSTFTPowerEncoder: STFT power front-end -> EnhancedPowerEncoder.  The front-end has no parameters; when
    the raw EEG asks for a gradient it is differentiated too (autograd.StftFrontEndFn)."""
    _need_gpu(x)
    enc = m.encoder
    if enc.training or _wants_grad(enc, x):
        from .autograd import PowerEncoderFn, StftFrontEndFn
        if x.requires_grad and torch.is_grad_enabled():           # d / d raw EEG: the front-end joins the tape
            spec = StftFrontEndFn.apply(x, m.n_ffts, m.hop, m.normalize)
        else:
            with torch.no_grad():
                spec = stft_front_end(x, m.n_ffts, m.hop, m.normalize)
        return PowerEncoderFn.run(enc, spec, packed=True)
    with torch.no_grad():
        return _power_forward_ntc(enc, stft_front_end(x, m.n_ffts, m.hop, m.normalize))


def power_encoder_forward(m, x):
    """EnhancedPowerEncoder (enhanced_models_v4.py:258-285)."""
    _need_gpu(x)
    if m.training or _wants_grad(m, x):
        from .autograd import PowerEncoderFn
        return PowerEncoderFn.run(m, x)
    with torch.no_grad():
        return _power_forward_ntc(m, pack_nct(x.float()))


class _Merged:
    """attribute bag standing in for an nn.Conv1d / nn.BatchNorm1d built from several modules"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def power_merge_call(mode: int, parts, merged, tracked=(None, None, None), cin: int = 0, ks=(3, 5, 7), cinp: int = 0):
    """mm_power_merge: ``parts`` = six triples (conv weight, conv bias, BN weight, BN bias, running mean, running var) of
    fp32 tensors or None, ``merged`` = the six merged tensors or None (include/mmeeg_hip.h: mm_power_merge_t)"""
    import ctypes
    import struct
    ptr = lambda t: 0 if t is None else t.data_ptr()      # noqa: E731
    flat = [ptr(t) for trip in parts for t in trip] + [ptr(t) for t in tracked] + [ptr(t) for t in merged]
    buf = struct.pack("<27Q6i", *flat, int(cin), *[int(k) for k in ks], int(cinp), 0)
    host = ctypes.create_string_buffer(buf, len(buf))
    _hip.call("mm_power_merge", ctypes.addressof(host), int(mode))


def _power_merged_train(m, need_dgrad: bool = True):
    """the three conv scales as ONE Conv1d(C -> 192, k=7, p=3) + BatchNorm1d(192) whose tensors are
    fresh leaves (requires_grad as the parts'), so the generic conv/BN forward+backward applies;
    autograd.power_encoder_bwd adds their gradients back into the six real parameters.  Built by ONE launch
    (mm_power_merge mode 0: the pads / cats of the three branches were ~20 tiny torch launches per step).
    ``need_dgrad`` False (no gradient w.r.t. the input: the trainers): the merged weight exists only as the forward
    kernel's bf16 image (mode 3), handed to the weight cache - its fp32 tensor is a never-read placeholder."""
    seqs = (m.conv_scale1, m.conv_scale2, m.conv_scale3)
    ref = seqs[0][0].weight
    cin = seqs[0][0].in_channels
    ks = tuple(s[0].kernel_size[0] for s in seqs)
    assert all(s[0].out_channels == 64 and s[0].in_channels == cin and s[0].weight.dtype == _F32 for s in seqs), "three 64-channel branches"
    w = _empty((192, cin, 7), _F32, ref)
    vec = [_empty((192,), _F32, ref) for _ in range(5)]
    parts = ([s[0].weight.detach() for s in seqs], [s[0].bias.detach() for s in seqs], [s[1].weight.detach() for s in seqs],
             [s[1].bias.detach() for s in seqs], [s[1].running_mean for s in seqs], [s[1].running_var for s in seqs])
    if need_dgrad:
        power_merge_call(0, parts, [w] + vec, cin=cin, ks=ks)
    else:
        cinp = cpad(cin)
        wf = _empty((192, 7, cinp), _BF, ref)
        power_merge_call(3, parts, [wf] + vec, cin=cin, ks=ks, cinp=cinp)
        weights.seed(w, wf, None, cinp, cpad(192))
    w._mm_transient = True        # rebuilt every step: a trainer's recorded weight list must not hold on to this one
    conv = _Merged(weight=w.requires_grad_(any(s[0].weight.requires_grad for s in seqs)),
                   bias=vec[0].requires_grad_(any(s[0].bias.requires_grad for s in seqs)),
                   kernel_size=(7,), padding=(3,), in_channels=cin, out_channels=192,
                   parts=[(s[0].weight, 64 * i) for i, s in enumerate(seqs)])        # (real weight, its first merged output channel)
    bn = _Merged(weight=vec[1].requires_grad_(any(s[1].weight.requires_grad for s in seqs)),
                 bias=vec[2].requires_grad_(any(s[1].bias.requires_grad for s in seqs)),
                 running_mean=vec[3], running_var=vec[4],
                 num_batches_tracked=None, num_features=192, eps=seqs[0][1].eps, momentum=seqs[0][1].momentum)
    assert all(s[1].eps == bn.eps and s[1].momentum == bn.momentum for s in seqs)
    return conv, bn, seqs


def _power_forward_impl(m, xb: torch.Tensor, training: bool, need_dgrad: bool, save=None):
    """EnhancedPowerEncoder on packed (B, T, Cp) bf16 input, train / frozen-BN forms."""
    save = training if save is None else save
    conv, bn, seqs = _power_merged_train(m, need_dgrad)
    r, s0 = conv_bn_act(xb, conv, bn, training=training, need_dgrad=need_dgrad, save=save)
    if training:                                  # running statistics live in the three real modules: handed back in one launch
        none3 = [None] * 3
        power_merge_call(1, (none3, none3, none3, none3, [sq[1].running_mean for sq in seqs], [sq[1].running_var for sq in seqs]),
                         [None, None, None, None, bn.running_mean, bn.running_var],
                         tracked=[sq[1].num_batches_tracked for sq in seqs], cin=conv.in_channels,
                         ks=[sq[0].kernel_size[0] for sq in seqs])
    T = xb.shape[1]
    p = m.drop_p if training else 0.0
    r, s1 = conv_bn_act(r["bf16"], m.fusion[0], m.fusion[1], training=training, drop_p=p,
                        pe=pe_table(m.pos_encoder, T), pe_drop_p=(m.pos_encoder.dropout.p if training else 0.0),
                        want_f32=True, want_bf16=False, need_dgrad=True, save=save)
    out, blocks, sh, h = _encoder_tail_impl(m, r["f32"], training, True, save)
    return out, dict(convs=[s0, s1], blocks=blocks, head=sh, merged=(conv, bn, seqs), tokens=h)


def _power_forward_ntc(m, xb):
    B, T, cp = xb.shape
    w192, o4 = _power_merged(m, xb.device)
    wf = _empty((192, 7, cp), _BF, xb)
    _hip.call("mm_prep_conv_weight", w192, wf, None, 192, w192.shape[1], 7, cp, 0)
    h = igemm(xb, wf, 7, 3, 192, scale=o4[0], shift=o4[1], act="gelu")["bf16"]
    r, _ = conv_bn_act(h, m.fusion[0], m.fusion[1], training=False, pe=pe_table(m.pos_encoder, T),
                       want_f32=True, want_bf16=False)
    tok = r["f32"]
    for blk in m.transformer_layers:
        tok, _ = transformer_block_fwd(tok, blk, False)
    out, _ = pooled_head_fwd(tok, m.output_proj[2])
    return out


def weighted_cross_entropy(logits, target, weight=None):
    """nn.CrossEntropyLoss(weight=...) on the HIP path (_test_bridge.py:858)."""
    _need_gpu(logits, target)
    from . import small_autograd as sa
    w = weight.float().contiguous() if weight is not None else None
    return sa.WeightedCEFn.apply(logits, target, w)


def focal_loss(logits, target, alpha=0.25, gamma=2.0, reduction="mean"):
    """FocalLoss of the EEG notebook (CrossModal_EEG_scr.ipynb cell 20) on the HIP path."""
    _need_gpu(logits, target)
    from . import small_autograd as sa
    return sa.FocalLossFn.apply(logits, target, float(alpha), float(gamma), reduction)


def drop_path(x, drop_prob):
    """training-mode stochastic depth (crossmodal_v4_enhancements.py:639-650) in one launch"""
    _need_gpu(x)
    from . import small_autograd as sa
    return sa.DropPathFn.apply(x, float(drop_prob))


def smoothed_cross_entropy(pred, target, smoothing):
    _need_gpu(pred, target)
    from . import small_autograd as sa
    return sa.SmoothedCEFn.apply(pred, target, float(smoothing))

"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional fp32 CPU restatement of the reference hot path.  Every function
takes a ``state_dict``-style mapping ``sd`` (key -> tensor, the reference's own
key names) plus a key ``prefix`` and recomputes the reference arithmetic with
plain torch ops.  ``train=True`` means "BatchNorm uses batch statistics";
dropout is never applied (parity is only defined for p=0 / eval mode because
the reference draws its masks from torch's global RNG stream).

All citations are file:line under /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]
_BN_EPS = 1e-5
_LN_EPS = 1e-5


# --------------------------------------------------------------------------
# leaf helpers
# --------------------------------------------------------------------------
def _lin(sd: SD, p: str, x):
    return F.linear(x, sd[p + "weight"], sd.get(p + "bias"))


def _bn(sd: SD, p: str, x, train: bool):
    """nn.BatchNorm{1,3}d arithmetic; batch stats when train (biased var)."""
    if train:
        dims = [0] + list(range(2, x.dim()))
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
    else:
        mean, var = sd[p + "running_mean"], sd[p + "running_var"]
    shape = [1, -1] + [1] * (x.dim() - 2)
    xh = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + _BN_EPS)
    return xh * sd[p + "weight"].view(shape) + sd[p + "bias"].view(shape)


def _ln(sd: SD, p: str, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], _LN_EPS)


def gelu(x):
    """exact erf GELU (nn.GELU() default; enhanced_models_v4.py:87,131)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def sinusoid_table(d_model: int, max_len: int = 5000):
    """enhanced_models_v4.py:37-41 -> (max_len, 1, d_model)."""
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32)
                    * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, 1, d_model)
    pe[:, 0, 0::2] = torch.sin(pos * div)
    pe[:, 0, 1::2] = torch.cos(pos * div)
    return pe


# --------------------------------------------------------------------------
# a1 PositionalEncoding  (enhanced_models_v4.py:44-55)
# --------------------------------------------------------------------------
def positional_encoding(sd: SD, p: str, x):
    pe = sd[p + "pe"]
    if x.dim() == 3 and x.size(1) != 1:          # batch-first (B, L, d)
        return x + pe[: x.size(1)].transpose(0, 1)
    return x + pe[: x.size(0)]                   # reference quirk for L == 1


# --------------------------------------------------------------------------
# nn.MultiheadAttention arithmetic (packed in_proj, batch_first)
# --------------------------------------------------------------------------
def multihead_attention(sd: SD, p: str, q_in, kv_in, nhead: int, attn_mask=None):
    """returns (out (B,Lq,E), head-averaged weights (B,Lq,Lk)).  ``attn_mask`` (Lq, Lk): boolean
    (True = not allowed) or additive float, as nn.MultiheadAttention takes it (enhanced_models_v4.py:98); its 3-D form
    (B * nhead, Lq, Lk) - matrix b * nhead + h for head h of sample b - is accepted too (the reference never passes it:
    tests/test_oracle_golden.py pins that branch against torch.nn.MultiheadAttention itself)."""
    E = q_in.shape[-1]
    W, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(q_in, W[:E], b[:E])
    k = F.linear(kv_in, W[E:2 * E], b[E:2 * E])
    v = F.linear(kv_in, W[2 * E:], b[2 * E:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    dh = E // nhead
    q = q.view(B, Lq, nhead, dh).transpose(1, 2)
    k = k.view(B, Lk, nhead, dh).transpose(1, 2)
    v = v.view(B, Lk, nhead, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if attn_mask is not None:
        if attn_mask.dim() == 3:
            attn_mask = attn_mask.view(B, nhead, Lq, Lk)
        if attn_mask.dtype == torch.bool:
            s = s.masked_fill(attn_mask, float("-inf"))
        else:
            s = s + attn_mask.to(s.dtype)
    a = torch.softmax(s, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, Lq, E)
    o = F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])
    return o, a.mean(dim=1)


# --------------------------------------------------------------------------
# a2 TemporalTransformerBlock (enhanced_models_v4.py:89-107), pre-norm
# --------------------------------------------------------------------------
def transformer_block(sd: SD, p: str, x, nhead: int, mask=None):
    h = _ln(sd, p + "norm1.", x)
    h, _ = multihead_attention(sd, p + "self_attn.", h, h, nhead, attn_mask=mask)
    x = x + h
    h = _ln(sd, p + "norm2.", x)
    h = _lin(sd, p + "linear2.", gelu(_lin(sd, p + "linear1.", h)))
    return x + h


def _num_layers(sd: SD, p: str) -> int:
    n = 0
    while (p + f"transformer_layers.{n}.norm1.weight") in sd:
        n += 1
    return n


def _encoder_tail(sd: SD, p: str, x_bcl, nhead: int, stages: Optional[dict]):
    """shared tail of a3/a4: transpose, PE, blocks, mean-pool, Linear, GELU."""
    x = positional_encoding(sd, p + "pos_encoder.", x_bcl.transpose(1, 2))
    if stages is not None:
        stages["pos"] = x
    for i in range(_num_layers(sd, p)):
        x = transformer_block(sd, p + f"transformer_layers.{i}.", x, nhead)
        if stages is not None:
            stages[f"block{i}"] = x
    pooled = x.mean(dim=1)                       # AdaptiveAvgPool1d(1)+Flatten
    return gelu(_lin(sd, p + "output_proj.2.", pooled))


# --------------------------------------------------------------------------
# a3 EnhancedERPEncoder (enhanced_models_v4.py:128-144, 169-193)
# --------------------------------------------------------------------------
def erp_encoder(sd: SD, x, p: str = "", nhead: int = 4, train: bool = False,
                stages: Optional[dict] = None):
    c = p + "conv_layers."
    h = gelu(_bn(sd, c + "1.", F.conv1d(x, sd[c + "0.weight"], sd[c + "0.bias"], padding=3), train))
    if stages is not None:
        stages["conv1"] = h
    h = gelu(_bn(sd, c + "5.", F.conv1d(h, sd[c + "4.weight"], sd[c + "4.bias"], padding=2), train))
    h = F.max_pool1d(h, 2)
    if stages is not None:
        stages["conv2"] = h
    h = gelu(_bn(sd, c + "10.", F.conv1d(h, sd[c + "9.weight"], sd[c + "9.bias"], padding=1), train))
    if stages is not None:
        stages["conv3"] = h
    return _encoder_tail(sd, p, h, nhead, stages)


# --------------------------------------------------------------------------
# a4 EnhancedPowerEncoder (enhanced_models_v4.py:210-234, 258-285)
# --------------------------------------------------------------------------
def power_encoder(sd: SD, x, p: str = "", nhead: int = 4, train: bool = False,
                  stages: Optional[dict] = None):
    branches = []
    for name, pad in (("conv_scale1.", 1), ("conv_scale2.", 2), ("conv_scale3.", 3)):
        q = p + name
        branches.append(gelu(_bn(sd, q + "1.", F.conv1d(x, sd[q + "0.weight"], sd[q + "0.bias"], padding=pad), train)))
    h = torch.cat(branches, dim=1)
    q = p + "fusion."
    h = gelu(_bn(sd, q + "1.", F.conv1d(h, sd[q + "0.weight"], sd[q + "0.bias"]), train))
    if stages is not None:
        stages["fusion"] = h
    return _encoder_tail(sd, p, h, nhead, stages)


# --------------------------------------------------------------------------
# a5 LearnedFusionModule (enhanced_models_v4.py:453-488)
# --------------------------------------------------------------------------
def learned_fusion(sd: SD, feats: Sequence[torch.Tensor], p: str = ""):
    tau = sd[p + "temperature"]
    static = torch.softmax(sd[p + "fusion_logits"] / tau, dim=0)
    g = gelu(_lin(sd, p + "gate_net.0.", torch.cat(list(feats), dim=1)))
    dyn = torch.softmax(_lin(sd, p + "gate_net.3.", g) / tau, dim=1)
    w = 0.5 * static.unsqueeze(0) + 0.5 * dyn
    fused = (torch.stack(list(feats), dim=1) * w.unsqueeze(2)).sum(dim=1)
    return fused, w


# --------------------------------------------------------------------------
# a6 Lite encoders (crossmodal_v4_enhancements.py:817-877)
# --------------------------------------------------------------------------
def _lite_encoder(sd: SD, x, p: str, pad1: int, pad2: int, train: bool):
    c = p + "conv_layers."
    h = gelu(_bn(sd, c + "1.", F.conv1d(x, sd[c + "0.weight"], sd[c + "0.bias"], padding=pad1), train))
    h = F.max_pool1d(h, 2)
    h = gelu(_bn(sd, c + "6.", F.conv1d(h, sd[c + "5.weight"], sd[c + "5.bias"], padding=pad2), train))
    return gelu(_lin(sd, p + "output.1.", h.mean(dim=2)))


def lite_erp_encoder(sd: SD, x, p: str = "", train: bool = False):
    return _lite_encoder(sd, x, p, 3, 2, train)      # k7 then k5


def lite_power_encoder(sd: SD, x, p: str = "", train: bool = False):
    return _lite_encoder(sd, x, p, 2, 1, train)      # k5 then k3


# --------------------------------------------------------------------------
# a7 EnhancedConnEncoder / HybridFusionModule / V4Lite
#    (crossmodal_v4_enhancements.py:695-739, 778-810, 920-944)
# --------------------------------------------------------------------------
def conn_encoder(sd: SD, x, p: str = "", train: bool = False):
    if x.dim() > 2:
        x = x.reshape(x.size(0), -1)
    h = gelu(_bn(sd, p + "proj1.1.", _lin(sd, p + "proj1.0.", x), train))
    h = gelu(_bn(sd, p + "proj2.1.", _lin(sd, p + "proj2.0.", h), train))
    gate = torch.sigmoid(_lin(sd, p + "attention.2.", torch.tanh(_lin(sd, p + "attention.0.", h))))
    h = h * gate
    return gelu(_bn(sd, p + "output.1.", _lin(sd, p + "output.0.", h), train))


def hybrid_fusion(sd: SD, erp, pw, conn, p: str = "", conn_boost: float = 1.3,
                  train: bool = False):
    g = gelu(_lin(sd, p + "erp_pw_gate.0.", torch.cat([erp, pw], dim=1)))
    gate = torch.softmax(_lin(sd, p + "erp_pw_gate.3.", g), dim=-1)
    mix = gate[:, 0:1] * erp + gate[:, 1:2] * pw
    comb = torch.cat([mix, conn * conn_boost], dim=1)
    fused = gelu(_bn(sd, p + "late_fusion.1.", _lin(sd, p + "late_fusion.0.", comb), train))
    return fused, gate


def trimodal_lite(sd: SD, erp, pw, conn, p: str = "", conn_boost: float = 1.3,
                  train: bool = False):
    """EnhancedTriModalFusionNetV4Lite.forward -> (logits, fused, gate)."""
    e = lite_erp_encoder(sd, erp, p + "erp_encoder.", train)
    w = lite_power_encoder(sd, pw, p + "pw_encoder.", train)
    c = conn_encoder(sd, conn, p + "conn_encoder.", train)
    fused, gate = hybrid_fusion(sd, e, w, c, p + "fusion.", conn_boost, train)
    h = gelu(_bn(sd, p + "classifier.1.", _lin(sd, p + "classifier.0.", fused), train))
    return _lin(sd, p + "classifier.4.", h), fused, gate


# --------------------------------------------------------------------------
# a8 loss / schedule (crossmodal_v4_enhancements.py:665-677, 1084-1112)
# --------------------------------------------------------------------------
def label_smoothing_ce(logits, target, smoothing: float = 0.1):
    lp = torch.log_softmax(logits, dim=-1)
    nll = -lp.gather(-1, target.unsqueeze(1)).squeeze(1)
    return ((1.0 - smoothing) * nll + smoothing * (-lp.mean(dim=-1))).mean()


def cosine_warmup_lr(epoch: int, base_lr: float, warmup: int, total: int,
                     min_lr: float = 1e-6) -> float:
    """lr after the ``epoch``-th call of CosineAnnealingWarmup.step()."""
    if epoch <= warmup:
        return base_lr * (epoch / warmup)
    prog = (epoch - warmup) / (total - warmup)
    return min_lr + 0.5 * (base_lr - min_lr) * (1 + math.cos(math.pi * prog))


# --------------------------------------------------------------------------
# a9 fMRI tabular encoders + fusion (fmri_utils.py:23-103)
# --------------------------------------------------------------------------
def _fmri_mlp(sd: SD, x, p: str, train: bool):
    h = torch.relu(_bn(sd, p + "encoder.1.", _lin(sd, p + "encoder.0.", x), train))
    return torch.relu(_bn(sd, p + "encoder.5.", _lin(sd, p + "encoder.4.", h), train))


def fmri_fusion_net(sd: SD, act, conn, p: str = "", task: str = "classification",
                    train: bool = False):
    a = _fmri_mlp(sd, act, p + "activation_encoder.", train)
    c = _fmri_mlp(sd, conn, p + "connectivity_encoder.", train)
    w = torch.softmax(torch.stack([sd[p + "activation_weight"], sd[p + "connectivity_weight"]]), dim=0)
    comb = torch.cat([a * w[0], c * w[1]], dim=1)
    fused = torch.relu(_bn(sd, p + "fusion.1.", _lin(sd, p + "fusion.0.", comb), train))
    out = _lin(sd, p + "head.3.", torch.relu(_lin(sd, p + "head.0.", fused)))
    if task == "regression":
        out = out.squeeze(-1)
    return out, fused


def fmri_single_branch(sd: SD, x, p: str = "", task: str = "classification", train: bool = False):
    """fMRIActivationOnly / fMRIConnectivityOnly (fMRI_CODE/run_fmri_v11.py:311-370): the tabular encoder MLP
    (:272-305, the same arithmetic as fmri_utils.py:23-56 which the a9 golden pins) -> Linear -> ReLU -> Dropout ->
    Linear.  run_fmri_v11.py itself is not importable here (seaborn), so these two classes have no golden of their
    own: they are pinned through their parts."""
    feat = _fmri_mlp(sd, x, p + "encoder.", train)
    out = _lin(sd, p + "head.3.", torch.relu(_lin(sd, p + "head.0.", feat)))
    return out.squeeze(-1) if task == "regression" else out


# --------------------------------------------------------------------------
# a11 EEGfMRIBridgeFusionNet (bridge_utils.py:68-103)
# --------------------------------------------------------------------------
def projection_head(sd: SD, p: str, x):
    """Linear -> LayerNorm -> GELU (bridge_utils.py:34-45), dropout off."""
    return gelu(_ln(sd, p + "1.", _lin(sd, p + "0.", x)))


def bridge_net(sd: SD, eeg, fmri, p: str = "", nhead: int = 4):
    ep = projection_head(sd, p + "eeg_proj.", eeg)
    fp = projection_head(sd, p + "fmri_proj.", fmri)
    seq = torch.stack([ep, fp], dim=1)
    att, attw = multihead_attention(sd, p + "cross_attn.", ep.unsqueeze(1), seq, nhead)
    fused, fw = learned_fusion(sd, [att.squeeze(1), fp], p + "fusion.")
    h = torch.relu(_ln(sd, p + "classifier.1.", _lin(sd, p + "classifier.0.", fused)))
    logits = _lin(sd, p + "classifier.4.", h)
    return logits, fused, fw, attw


# --------------------------------------------------------------------------
# full V4 classifiers composed from a3 / a4 / a5
# (crossmodal_v4_enhancements.py:278-388 tri-modal, :403-466 bi-directional
# cross attention, :473-570 bi-modal).  Dropout is the identity here (eval, or
# train with p = 0); ``train`` switches BatchNorm to batch statistics.
# --------------------------------------------------------------------------
def _mlp_bn_gelu(sd: SD, p: str, x, i_lin: int, train: bool):
    return gelu(_bn(sd, f"{p}{i_lin + 1}.", _lin(sd, f"{p}{i_lin}.", x), train))


def _v4_classifier(sd: SD, p: str, fused, train: bool):
    h = _mlp_bn_gelu(sd, p, fused, 0, train)
    h = _mlp_bn_gelu(sd, p, h, 4, train)
    return _lin(sd, p + "8.", h)


def trimodal_v4(sd: SD, erp, pw, conn, p: str = "", nhead: int = 4, train: bool = False):
    """-> (logits, fusion weights (B, 3), fused (B, H))"""
    e = erp_encoder(sd, erp, p + "erp_encoder.", nhead, train)
    w = power_encoder(sd, pw, p + "pw_encoder.", nhead, train)
    c = conn.reshape(conn.shape[0], -1)
    c = _mlp_bn_gelu(sd, p + "conn_encoder.", c, 0, train)
    c = _mlp_bn_gelu(sd, p + "conn_encoder.", c, 4, train)
    stack = torch.stack([e, w, c], dim=1)
    enh, _ = multihead_attention(sd, p + "cross_attn.", e.unsqueeze(1), stack, nhead)
    fused, weights = learned_fusion(sd, [enh.squeeze(1), w, c], p + "fusion.")
    return _v4_classifier(sd, p + "classifier.", fused, train), weights, fused


def bidirectional_cross_attention(sd: SD, e, w, p: str = "", nhead: int = 4):
    comb = torch.stack([e, w], dim=1)
    ea, _ = multihead_attention(sd, p + "erp_to_pw_attn.", e.unsqueeze(1), comb, nhead)
    wa, _ = multihead_attention(sd, p + "pw_to_erp_attn.", w.unsqueeze(1), comb, nhead)
    ea, wa = ea.squeeze(1), wa.squeeze(1)
    ge = torch.sigmoid(_lin(sd, p + "erp_gate.0.", torch.cat([e, ea], dim=1)))
    gw = torch.sigmoid(_lin(sd, p + "pw_gate.0.", torch.cat([w, wa], dim=1)))
    return _ln(sd, p + "norm_erp.", e + ge * ea), _ln(sd, p + "norm_pw.", w + gw * wa)


def smart_fusion_v4(sd: SD, erp, pw, p: str = "", nhead: int = 4, train: bool = False,
                    use_cross_attention: bool = True):
    """-> (logits, fusion weights (B, 2), fused (B, H))"""
    e = erp_encoder(sd, erp, p + "erp_encoder.", nhead, train)
    w = power_encoder(sd, pw, p + "pw_encoder.", nhead, train)
    if use_cross_attention:
        e, w = bidirectional_cross_attention(sd, e, w, p + "cross_attention.", nhead)
    fused, weights = learned_fusion(sd, [e, w], p + "fusion.")
    return _v4_classifier(sd, p + "classifier.", fused, train), weights, fused


# ==========================================================================
# EXTENSIONS named by north_star, absent from the reference
# ("parity unpinned by reference"; definitions in DESIGN.md)
# ==========================================================================
def max_pool3d_routed(h, route):
    """MaxPool3d(2) with the window member to take GIVEN (``route`` (B, C, D/2, H/2, W/2) int64, member index
    (dd << 2) | (hh << 1) | ww): what max-pooling computes whenever ``route`` is its own arg-max.  The parity tests pass
    the routing the HIP path took, so that a flip between two near-equal window members (bf16 operands) is not counted
    as a gradient error of everything below the pool."""
    B, C, D, H, W = h.shape
    win = h.reshape(B, C, D // 2, 2, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(B, C, D // 2, H // 2, W // 2, 8)
    return win.gather(-1, route.unsqueeze(-1)).squeeze(-1)


def volume_encoder3d(sd: SD, x, p: str = "", train: bool = False,
                     stages: Optional[dict] = None, route=None):
    """a-X1: Conv3d(1->32)-BN-GELU-MaxPool2 / Conv3d(32->64)-BN-GELU-MaxPool2 /
    Conv3d(64->128)-BN-GELU / global-avg-pool / Linear(128->64)-GELU.
    ``route`` = (route1, route2): evaluate the two max-pools with given window members (`max_pool3d_routed`)."""
    c = p + "conv_layers."
    h = gelu(_bn(sd, c + "1.", F.conv3d(x, sd[c + "0.weight"], sd[c + "0.bias"], padding=1), train))
    if stages is not None:
        stages["act1"] = h                                  # (before the pool)
    h = F.max_pool3d(h, 2) if route is None else max_pool3d_routed(h, route[0])
    if stages is not None:
        stages["conv1"] = h
    h = gelu(_bn(sd, c + "6.", F.conv3d(h, sd[c + "5.weight"], sd[c + "5.bias"], padding=1), train))
    if stages is not None:
        stages["act2"] = h
    h = F.max_pool3d(h, 2) if route is None else max_pool3d_routed(h, route[1])
    if stages is not None:
        stages["conv2"] = h
    h = gelu(_bn(sd, c + "11.", F.conv3d(h, sd[c + "10.weight"], sd[c + "10.bias"], padding=1), train))
    if stages is not None:
        stages["conv3"] = h
    return gelu(_lin(sd, p + "output_proj.2.", h.mean(dim=(2, 3, 4))))


def l2_normalize(x, eps: float = 1e-12):
    return x / x.norm(dim=1, keepdim=True).clamp_min(eps)


def contrastive_head(sd: SD, eeg_feat, fmri_feat, p: str = ""):
    """a-X2 embeddings: reuse the bridge projection heads, then L2-normalise."""
    ze = l2_normalize(projection_head(sd, p + "eeg_proj.", eeg_feat))
    zf = l2_normalize(projection_head(sd, p + "fmri_proj.", fmri_feat))
    return ze, zf


def clip_loss(ze, zf_all, ze_all, zf, logit_scale, row0: int = 0):
    """a-X2 symmetric InfoNCE for local rows [row0, row0+B) against the global
    (all-gathered) columns.  Returns (loss, top1_e2f, top1_f2e, S_e2f)."""
    B = ze.shape[0]
    tgt = torch.arange(row0, row0 + B)
    s_ef = logit_scale * ze @ zf_all.t()
    s_fe = logit_scale * zf @ ze_all.t()
    loss = 0.5 * (F.cross_entropy(s_ef, tgt) + F.cross_entropy(s_fe, tgt))
    acc_e = (s_ef.argmax(1) == tgt).float().mean()
    acc_f = (s_fe.argmax(1) == tgt).float().mean()
    return loss, acc_e, acc_f, s_ef


def normalize_modality(feat, eps: float = 1e-8):
    """run_training_lite.py:48-51: z-score over ALL elements of one sample's feature array, here batched: feat (B, ...).
    The reference applies it to NUMPY arrays (:108, 162, 216: ``mat[key].astype(np.float32)``), whose ``.std()`` is the
    population standard deviation (ddof = 0) - not torch's unbiased default."""
    flat = feat.flatten(1)
    mean = flat.mean(dim=1).view(-1, *([1] * (feat.dim() - 1)))
    std = flat.std(dim=1, unbiased=False).view(-1, *([1] * (feat.dim() - 1))) + eps
    return (feat - mean) / std


def stft_power_encoder(sd: SD, x, n_ffts=(64, 128), hop: int = 32, p: str = "encoder.", nhead: int = 4,
                       normalize: bool = True, train: bool = False):
    """a-X3 + a4: multi-scale STFT power (channel-concatenated) [-> per-sample z-score, the reference's
    normalize_modality on its power features, run_training_lite.py:162] -> EnhancedPowerEncoder."""
    spec = torch.cat([stft_power(x, n, hop) for n in n_ffts], dim=1)
    if normalize:
        spec = normalize_modality(spec)
    return power_encoder(sd, spec, p, nhead, train=train)


def stft_power(x, n_fft: int, hop: int):
    """a-X3: per-channel Hann STFT power, (B,C,T) -> (B, C*F, frames)."""
    B, C, T = x.shape
    win = torch.hann_window(n_fft, periodic=True)
    z = torch.stft(x.reshape(B * C, T), n_fft, hop_length=hop, window=win,
                   center=True, pad_mode="reflect", return_complex=True)
    pw = (z.real ** 2 + z.imag ** 2)
    return pw.reshape(B, C * pw.shape[1], pw.shape[2])

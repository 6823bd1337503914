"""TEST INFRASTRUCTURE ONLY.  Host replica of the HIP path's counter-hash dropout masks
(csrc/common.h: mm_hash / dropout_scale, csrc/attention.hip: attn_block_hash / attn_keep2) and a restatement of the
reference's TRAIN-mode EnhancedERPEncoder forward (enhanced_models_v4.py:128-144, 44-55, 88-105,
161-193) with an explicit keep-mask at every nn.Dropout site, so that a whole-encoder forward/backward
with dropout > 0 can be checked against the CPU oracle.  The reference draws its masks from torch's RNG
stream, which no kernel can reproduce; what is pinned here is that every dropout site exists, sits at
the right place in the graph, and that backward differentiates the same masked function.

Mask element index = flat index of the dropped tensor in the kernels' channels-last layout:
conv stages (B, T, C), token stages (B*L, d), attention probabilities (B*H, L, L), head (B, H)."""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as TF

from . import ref_functional as RF


def keep_scale(seed: int, n: int, p: float) -> torch.Tensor:
    """[n] float32: 1/(1-p) where hash(seed, i) >= p * 2^32, else 0 (common.h: dropout_scale)."""
    if p <= 0.0:
        return torch.ones(n)
    idx = torch.arange(n, dtype=torch.int64)
    x = (idx * 0x9E3779B1 + int(seed)) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 13
    return (x >= int(p * 4294967296.0)).float() / (1.0 - p)


def attn_keep_scale(seed: int, bh: int, L: int, p: float) -> torch.Tensor:
    """[bh, L, L] float32 attention-probability mask (attention.hip: attn_block_hash / attn_keep2): one hash per 2 x 2
    block (queries 2i, 2i + 1) x (keys 2j, 2j + 1), block index (bh * ceil(L / 2) + q // 2) * ceil(L / 2) + key // 2;
    byte 2 (q & 1) + (key & 1) of the word decides the score, keep iff byte >= t = round(p * 256); kept scores are scaled
    by 256 / (256 - t) (unbiased for the quantised probability)."""
    t8 = int(p * 256.0 + 0.5)
    if t8 == 0:
        return torch.ones(bh, L, L)
    Lh = (L + 1) // 2
    idx = torch.arange(bh * Lh * Lh, dtype=torch.int64)
    x = (idx * 0x9E3779B1 + int(seed)) & 0xFFFFFFFF
    x ^= x >> 13
    x = ((x & 0xFFFFFF) * 0xB5297B) & 0xFFFFFFFF
    x ^= x >> 15
    by = torch.stack([(x >> (8 * k)) & 0xFF for k in range(4)], dim=-1)                 # [blocks, 2 (q & 1) + (key & 1)]
    keep = (by >= t8).view(bh, Lh, Lh, 2, 2).permute(0, 1, 3, 2, 4).reshape(bh, 2 * Lh, 2 * Lh)[:, :L, :L]
    return keep.float() * (256.0 / (256.0 - t8))


def erp_encoder_train_with_masks(sd: Dict[str, torch.Tensor], x, seeds: List[int], p: float, p_attn: float,
                                 p_pe: float, nhead: int = 4, pre: str = ""):
    """``seeds``: the dropout seeds in the order the product path draws them (ops._next_seed):
    conv1, conv2, conv3, positional, then per block (attention probs, dropout1, FFN activation,
    dropout2), then the output head."""
    F = RF.F                                             # honours oracle.bf16_emulation.bf16_operands()
    it = iter(seeds)
    c = pre + "conv_layers."

    def drop_bct(h, prob):                               # (B, C, T) tensor, mask indexed as (B, T, C)
        B, C, T = h.shape
        m = keep_scale(next(it), B * T * C, prob).view(B, T, C).transpose(1, 2)
        return h * m

    h = RF.gelu(RF._bn(sd, c + "1.", F.conv1d(x, sd[c + "0.weight"], sd[c + "0.bias"], padding=3), True))
    h = drop_bct(h, p)
    h = RF.gelu(RF._bn(sd, c + "5.", F.conv1d(h, sd[c + "4.weight"], sd[c + "4.bias"], padding=2), True))
    h = drop_bct(TF.max_pool1d(h, 2), p)
    h = RF.gelu(RF._bn(sd, c + "10.", F.conv1d(h, sd[c + "9.weight"], sd[c + "9.bias"], padding=1), True))
    h = drop_bct(h, p)
    t = RF.positional_encoding(sd, pre + "pos_encoder.", h.transpose(1, 2))
    B, L, D = t.shape
    t = t * keep_scale(next(it), B * L * D, p_pe).view(B, L, D)
    for i in range(RF._num_layers(sd, pre)):
        q = pre + f"transformer_layers.{i}."
        sa, s1, s2, s3 = next(it), next(it), next(it), next(it)
        hn = RF._ln(sd, q + "norm1.", t)
        W, b = sd[q + "self_attn.in_proj_weight"], sd[q + "self_attn.in_proj_bias"]
        qkv = F.linear(hn, W, b)
        dh = D // nhead
        qh, kh, vh = (u.view(B, L, nhead, dh).transpose(1, 2) for u in qkv.split(D, dim=2))
        a = torch.softmax((qh @ kh.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
        a = a * attn_keep_scale(sa, B * nhead, L, p_attn).view(B, nhead, L, L)
        o = (a @ vh).transpose(1, 2).reshape(B, L, D)
        o = F.linear(o, sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"])
        t = t + o * keep_scale(s1, B * L * D, p).view(B, L, D)
        hn = RF._ln(sd, q + "norm2.", t)
        f = RF.gelu(F.linear(hn, sd[q + "linear1.weight"], sd[q + "linear1.bias"]))
        f = f * keep_scale(s2, f.numel(), p).view(f.shape)
        f = F.linear(f, sd[q + "linear2.weight"], sd[q + "linear2.bias"])
        t = t + f * keep_scale(s3, B * L * D, p).view(B, L, D)
    out = RF.gelu(F.linear(t.mean(dim=1), sd[pre + "output_proj.2.weight"], sd[pre + "output_proj.2.bias"]))
    return out * keep_scale(next(it), out.numel(), p).view(out.shape)


def volume_encoder_train_with_masks(sd: Dict[str, torch.Tensor], x, seeds: List[int], p: float, pre: str = ""):
    """TRAIN-mode fMRIVolumeEncoder3D (the a-X1 extension: Conv3d-BN-GELU-MaxPool-Dropout x2, Conv3d-BN-GELU-Dropout,
    global average, Linear-GELU-Dropout) with an explicit keep-mask at its four nn.Dropout sites.  ``seeds``: layer 1,
    layer 2, layer 3, output head - the order ops._vol_forward_impl draws them.  Mask index = flat index of the dropped
    tensor in the kernels' channels-last layout (B, D, H, W, C) - for the pooled layers the POOLED tensor."""
    F = RF.F
    it = iter(seeds)
    c = pre + "conv_layers."

    def drop_cl(h):
        B, C, D, H, W = h.shape
        m = keep_scale(next(it), h.numel(), p).view(B, D, H, W, C).permute(0, 4, 1, 2, 3)
        return h * m

    h = RF.gelu(RF._bn(sd, c + "1.", F.conv3d(x, sd[c + "0.weight"], sd[c + "0.bias"], padding=1), True))
    h = drop_cl(TF.max_pool3d(h, 2))
    h = RF.gelu(RF._bn(sd, c + "6.", F.conv3d(h, sd[c + "5.weight"], sd[c + "5.bias"], padding=1), True))
    h = drop_cl(TF.max_pool3d(h, 2))
    h = RF.gelu(RF._bn(sd, c + "11.", F.conv3d(h, sd[c + "10.weight"], sd[c + "10.bias"], padding=1), True))
    h = drop_cl(h)
    out = RF.gelu(F.linear(h.mean(dim=(2, 3, 4)), sd[pre + "output_proj.2.weight"], sd[pre + "output_proj.2.bias"]))
    return out * keep_scale(next(it), out.numel(), p).view(out.shape)


def contrastive_head_with_masks(sd: Dict[str, torch.Tensor], eeg_feat, fmri_feat, seeds: List[int], p: float, pre: str = ""):
    """both projection heads in TRAIN mode (bridge_utils.py:34-45: Linear -> LayerNorm -> GELU -> Dropout) followed by
    F.normalize; ``seeds`` = (EEG head, fMRI head), mask index b * N + n."""
    def head(q, x, seed):
        a = RF.gelu(RF._ln(sd, q + "1.", RF._lin(sd, q + "0.", x)))
        return RF.l2_normalize(a * keep_scale(seed, a.numel(), p).view(a.shape))
    return head(pre + "eeg_proj.", eeg_feat, seeds[0]), head(pre + "fmri_proj.", fmri_feat, seeds[1])


def bridge_step_with_masks(sd: Dict[str, torch.Tensor], eeg, fmri, seeds: List[int], p: float):
    """the whole contrastive training step's forward (what bench.py times, at dropout p) with every keep-mask explicit:
    names prefixed e. / f. / h. as in tests/test_trainer_gpu.py::_oracle_step; 13 + 4 + 2 seeds in draw order."""
    assert len(seeds) == 19, len(seeds)
    fe = erp_encoder_train_with_masks(sd, eeg, seeds[:13], p, p, p, pre="e.")
    ff = volume_encoder_train_with_masks(sd, fmri, seeds[13:17], p, pre="f.")
    ze, zf = contrastive_head_with_masks(sd, fe, ff, seeds[17:19], p, pre="h.bridge.")
    loss = RF.clip_loss(ze, zf, ze, zf, sd["h.logit_scale"].exp())[0]
    return loss, ze, zf

"""GPU: kernel-level parity of the C-ABI entry points against plain fp32 torch
on the CPU (operands pre-rounded to bf16 where the kernel's MFMA operands are
bf16, so the only remaining difference is fp32 accumulation order)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


def _hip():
    from multimodal_eeg_fmri_amd import _hip
    _hip.load()
    return _hip


def _prep_w(hip, w, cinp, coutp=None):
    cout, cin, k = w.shape
    wf = torch.empty(cout, k, cinp, dtype=torch.bfloat16, device="cuda")
    wd = None
    if coutp:
        wd = torch.empty(cinp, k, coutp, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_prep_conv_weight", w.cuda().contiguous(), wf, wd, cout, cin, k, cinp, coutp or 0)
    return wf, wd


def _cpad(C):
    cp = max(16, (C + 15) // 16 * 16)
    return cp if cp <= 32 else (cp + 31) // 32 * 32


@pytest.mark.parametrize("B,C,T,Cout,k", [(2, 64, 256, 64, 7), (3, 64, 200, 128, 5), (2, 128, 128, 128, 3),
                                          (8, 8, 256, 64, 7), (1, 128, 96, 384, 1), (2, 512, 64, 128, 1),
                                          (1, 16, 40, 48, 7)])
def test_conv1d_fwd_matches_torch(B, C, T, Cout, k):
    hip = _hip()
    g = torch.Generator().manual_seed(B * 1000 + C + T + Cout + k)
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(Cout, C, k, generator=g) / math.sqrt(C * k)
    bias = torch.randn(Cout, generator=g)
    cp = _cpad(C)
    xg = torch.empty(B, T, cp, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pack_nct_bf16", x.cuda(), xg, B, C, T, cp)
    want_pack = torch.zeros(B, T, cp)
    want_pack[:, :, :C] = _bf(x).transpose(1, 2)
    assert torch.equal(xg.float().cpu(), want_pack)
    wf, _ = _prep_w(hip, w, cp)
    out = torch.empty(B, T, Cout, dtype=torch.float32, device="cuda")
    hip.call("mm_conv1d_fwd", xg, wf, B, T, cp, Cout, k, k // 2, None, bias.cuda(), 0, None, None, 1,
             None, out, None, None, 0.0, 0)
    want = F.conv1d(_bf(x), _bf(w), bias, padding=k // 2).transpose(1, 2)
    torch.testing.assert_close(out.cpu(), want, rtol=1e-4, atol=1e-4)


def test_conv1d_fwd_epilogue_bn_gelu_pool_stats():
    hip = _hip()
    g = torch.Generator().manual_seed(5)
    B, C, T, Cout, k = 2, 64, 128, 128, 5
    x = torch.randn(B, C, T, generator=g)
    w = torch.randn(Cout, C, k, generator=g) / math.sqrt(C * k)
    scale = 0.5 + torch.rand(Cout, generator=g)
    shift = torch.randn(Cout, generator=g) * 0.3
    xg = torch.empty(B, T, C, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_pack_nct_bf16", x.cuda(), xg, B, C, T, C)
    wf, _ = _prep_w(hip, w, C)
    out = torch.empty(B, T // 2, Cout, dtype=torch.bfloat16, device="cuda")
    stats = torch.zeros(2, Cout, device="cuda")
    hip.call("mm_conv1d_fwd", xg, wf, B, T, C, Cout, k, k // 2, scale.cuda(), shift.cuda(), 1, None, None, 2,
             stats, None, out, None, 0.0, 0)
    z = F.conv1d(_bf(x), _bf(w), None, padding=k // 2) * scale[None, :, None] + shift[None, :, None]
    want = F.max_pool1d(F.gelu(z), 2).transpose(1, 2)
    torch.testing.assert_close(out.float().cpu(), want, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(stats[0].cpu(), z.sum(dim=(0, 2)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(stats[1].cpu(), (z * z).sum(dim=(0, 2)), rtol=1e-3, atol=1e-2)

"""stand-alone timing (graph-replayed) of the fused data-gradient + BatchNorm-reduce launch against its two parts"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_eeg_fmri_amd import _hip as hip
hip.load()
from kbench import graph_time

def case(B, T, Cin, Cout, k, pool, p):
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cin, Cout, k, generator=g) / math.sqrt(Cout * k)
    wf = torch.empty(Cin, k, Cout, dtype=torch.bfloat16, device="cuda")
    wd = torch.empty(Cout, k, Cin, dtype=torch.bfloat16, device="cuda")
    hip.call("mm_prep_conv_weight", w.cuda().contiguous(), wf, wd, Cin, Cout, k, Cout, Cin)
    dy = (torch.randn(B, T, Cin, generator=g) * 0.1).cuda().to(torch.bfloat16)
    yb = (torch.randn(B, T * pool, Cout, generator=g) * 1.2 + 0.1).cuda()
    out4 = torch.stack([0.5 + torch.rand(Cout, generator=g), torch.randn(Cout, generator=g) * 0.2,
                        torch.randn(Cout, generator=g) * 0.1, 0.8 + 0.4 * torch.rand(Cout, generator=g)]).cuda().contiguous()
    dx = torch.empty(B, T, Cout, dtype=torch.bfloat16, device="cuda")
    sums = torch.zeros(32, 2, Cout, device="cuda")
    t_d = graph_time(lambda: hip.call("mm_conv1d_fwd", dy, wd, B, T, Cin, Cout, k, k - 1 - k // 2, None, None, 0, None, None, 1,
                                      None, None, dx, None, 0.0, 0, None, None, 0))
    t_r = graph_time(lambda: hip.call("mm_bn_act_bwd_reduce", yb, out4, dx, None, sums, B, T * pool, Cout, 1, pool, 0, p, 7, 0.0, 0, None))
    t_f = graph_time(lambda: hip.call("mm_conv1d_dgrad_bn_reduce", dy, wd, B, T, Cin, Cout, k, k - 1 - k // 2, dx, yb, out4, sums,
                                      1, pool, 0, p, 7, None))
    print(f"B={B} T={T} {Cin}->{Cout} k={k} pool={pool}: dgrad {t_d:5.1f} us + reduce {t_r:5.1f} us = {t_d + t_r:5.1f} us; fused {t_f:5.1f} us")

case(32, 512, 128, 128, 3, 2, 0.3)
case(32, 1024, 128, 64, 5, 1, 0.3)

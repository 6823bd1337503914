import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import ref_functional as RF
from oracle.fixtures import build, seeded_randn
import multimodal_eeg_fmri_amd.fmri_utils as Fm
from multimodal_eeg_fmri_amd import ops
m = build(Fm.fMRIVolumeEncoder3D, 32, dropout=0.0).train()
x = seeded_randn(132, 4, 1, 16, 16, 16)
st = {}
with torch.no_grad():
    want = RF.volume_encoder3d(m.state_dict(), x, train=True, stages=st)
mg = m.cuda()
cl = mg.conv_layers
with torch.no_grad():
    h, s = ops.conv3d_l1_bn_act(x.cuda(), cl[0], cl[1], training=True, drop_p=0.0)
    w1 = st["conv1"].permute(0, 2, 3, 4, 1)
    print("L1 rel", ((h.float().cpu() - w1).norm() / w1.norm()).item())
    h2, s2 = ops.conv3d_bn_act(h, cl[5], cl[6], pool=True, training=True, drop_p=0.0, need_dgrad=False)
    w2 = st["conv2"].permute(0, 2, 3, 4, 1)
    print("L2 rel", ((h2.float().cpu() - w2).norm() / w2.norm()).item())
    h3, s3 = ops.conv3d_bn_act(h2, cl[10], cl[11], pool=False, training=True, drop_p=0.0, need_dgrad=False)
    w3 = st["conv3"].permute(0, 2, 3, 4, 1).reshape(4, -1, 128)
    print("L3 rel", ((h3.float().cpu() - w3).norm() / w3.norm()).item())
m2 = build(Fm.fMRIVolumeEncoder3D, 32, dropout=0.0).train().cuda()
y = m2(x.cuda())
print("full train fwd rel", ((y.detach().cpu() - want).norm() / want.norm()).item())
y2, sv = ops._vol_forward_impl(m2, x.cuda(), True, True)
print("impl rel", ((y2.detach().cpu() - want).norm() / want.norm()).item())
pooled = sv["head"]["pooled"].float().cpu()
wp = st["conv3"].mean(dim=(2, 3, 4))
print("pooled rel", ((pooled - wp).norm() / wp.norm()).item())

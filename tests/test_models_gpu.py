"""GPU: the product classes (HIP path) against the reference goldens and the
CPU oracle on identical seeded inputs.  Tolerances are stated per test:
bf16 MFMA operands with fp32 accumulation give ~1e-2 relative error per
activation, and the north-star bar of cosine similarity >= 1 - 1e-4 on the
embeddings."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_functional as RF
from oracle.fixtures import build, checksum, seeded_randn

import multimodal_eeg_fmri_amd.enhanced_models_v4 as E

pytestmark = pytest.mark.gpu
COS_TOL = 1e-4


def cos_min(a, b):
    return F.cosine_similarity(a.double().flatten(1), b.double().flatten(1), dim=1).min().item()


def rel_err(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("tag", ["c1", "c2"])
def test_a3_erp_encoder_eval_vs_golden(golden, tag):
    fx = golden(f"a3_erp_{tag}.npz")
    B, C, T = (int(v) for v in fx["shape"])
    m = build(E.EnhancedERPEncoder, int(fx["seed"]), C, 128, 2, 4, 0.3).eval()
    np.testing.assert_allclose(checksum(m), fx["cks"], rtol=1e-6, atol=1e-6)
    x = seeded_randn(int(fx["x_seed"]), B, C, T)
    m = m.cuda()
    with torch.no_grad():
        y = m(x.cuda()).cpu()
    want = torch.as_tensor(fx["out"])
    assert y.shape == want.shape and y.dtype == torch.float32
    c = cos_min(y, want)
    assert c >= 1 - COS_TOL, f"cosine {c}"
    assert rel_err(y, want) < 2e-2

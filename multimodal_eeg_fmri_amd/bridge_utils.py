"""EEG<->fMRI bridge on the MI355X HIP path.

``EEGfMRIBridgeFusionNet`` / ``BridgeFeatureDataset`` mirror the reference's
``bridge_utils.py:22-152`` (names, signatures, return conventions, state_dict).
``EEGfMRIContrastiveBridge`` is the north-star extension (SURVEY.md §8 a-X2):
the same two projection heads, L2-normalised, scored with a batch-pairwise
cosine-similarity matrix against all-gathered global negatives and trained with
a symmetric InfoNCE loss.  The reference trains a 2-class CE classifier and has
no similarity matrix, so that part's parity is unpinned by the reference.
"""
from __future__ import annotations

import logging
import math

import torch
import torch.nn as nn
from torch.utils.data import Dataset

from . import ops
from .enhanced_models_v4 import LearnedFusionModule

logger = logging.getLogger(__name__)


def _proj_head(in_dim, out_dim, dropout):
    return nn.Sequential(nn.Linear(in_dim, out_dim), nn.LayerNorm(out_dim),
                         nn.GELU(), nn.Dropout(dropout))


class EEGfMRIBridgeFusionNet(nn.Module):
    """eeg (B,128) + fmri (B,64) -> logits (B,2) [, fused, fusion_w, attn_w]."""

    def __init__(self, eeg_dim=128, fmri_dim=64, bridge_dim=128,
                 num_classes=2, num_heads=4, dropout=0.3):
        super().__init__()
        self.bridge_dim = bridge_dim
        self.eeg_proj = _proj_head(eeg_dim, bridge_dim, dropout)
        self.fmri_proj = _proj_head(fmri_dim, bridge_dim, dropout)
        self.cross_attn = nn.MultiheadAttention(bridge_dim, num_heads=num_heads,
                                                dropout=dropout, batch_first=True)
        self.fusion = LearnedFusionModule(num_modalities=2, hidden_dim=bridge_dim,
                                          use_temperature=True)
        self.classifier = nn.Sequential(
            nn.Linear(bridge_dim, bridge_dim // 2), nn.LayerNorm(bridge_dim // 2),
            nn.ReLU(), nn.Dropout(dropout), nn.Linear(bridge_dim // 2, num_classes))
        self.num_heads = num_heads
        self.drop_p = dropout

    def forward(self, eeg_feats, fmri_feats, return_features=False, return_weights=False):
        logits, fused, fusion_w, attn_w = ops.bridge_forward(self, eeg_feats, fmri_feats)
        out = [logits]
        if return_features:
            out.append(fused)
        if return_weights:
            out += [fusion_w, attn_w]
        return out[0] if len(out) == 1 else tuple(out)

    def get_fusion_weights(self):
        with torch.no_grad():
            temp = self.fusion.temperature
            w = torch.softmax(self.fusion.fusion_logits / temp, dim=0)
            return {"eeg_weight": w[0].item(), "fmri_weight": w[1].item(),
                    "temperature": temp.item()}


class EEGfMRIContrastiveBridge(nn.Module):
    """Contrastive projection bridge (extension; SURVEY.md §8 a-X2).

    ``bridge`` carries the reference-compatible parameters (its ``eeg_proj`` and
    ``fmri_proj`` are the trained heads); ``logit_scale`` = ln(1/tau) is the one
    extra learnable scalar (CLIP convention, tau0 = 0.07).
    """

    def __init__(self, eeg_dim=128, fmri_dim=64, bridge_dim=128, dropout=0.3,
                 init_tau: float = 0.07):
        super().__init__()
        self.bridge = EEGfMRIBridgeFusionNet(eeg_dim, fmri_dim, bridge_dim, dropout=dropout)
        self.logit_scale = nn.Parameter(torch.tensor(math.log(1.0 / init_tau)))

    def embed(self, eeg_feats, fmri_feats):
        """L2-normalised (ze, zf), each (B, bridge_dim) fp32."""
        return ops.contrastive_embed(self.bridge, eeg_feats, fmri_feats, self.training)

    def forward(self, eeg_feats, fmri_feats, group=None):
        """-> (loss, top1_eeg_to_fmri, top1_fmri_to_eeg).  With a process
        ``group`` the columns are the all-gathered global batch."""
        ze, zf = self.embed(eeg_feats, fmri_feats)
        return ops.clip_loss(ze, zf, self.logit_scale, group)


class BridgeFeatureDataset(Dataset):
    """Aligns dict-of-tensor EEG/fMRI features and labels on ``int(subject)``."""

    def __init__(self, eeg_features, fmri_features, labels, subject_list):
        eeg = {int(k): v for k, v in eeg_features.items()}
        fmri = {int(k): v for k, v in fmri_features.items()}
        lab = {int(k): v for k, v in labels.items()}
        self.samples = [
            {"eeg": eeg[s], "fmri": fmri[s], "label": lab[s], "subject": s}
            for s in sorted(int(x) for x in subject_list)
            if s in eeg and s in fmri and s in lab]
        if not self.samples:
            logger.error("!!! NO SAMPLES ALIGNED !!! Check subject IDs in EEG and fMRI feature dicts.")
        else:
            logger.info("BridgeFeatureDataset: %d aligned samples found.", len(self.samples))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        s = self.samples[idx]
        return s["eeg"], s["fmri"], s["label"], s["subject"]


class WeightedCrossEntropy(nn.Module):
    """``nn.CrossEntropyLoss(weight=class_w)`` of the bridge loop (_test_bridge.py:858) as one kernel."""

    def __init__(self, weight=None):
        super().__init__()
        self.register_buffer("weight", weight if weight is None else weight.clone().float())

    def forward(self, logits, target):
        return ops.weighted_cross_entropy(logits, target, self.weight)


def train_bridge_epoch(model, loader, optimizer, criterion, device, grad_clip=1.0):
    """one epoch of the reference bridge loop (_test_bridge.py:775-788).  With a
    ``multimodal_eeg_fmri_amd.optim.FusedAdamW`` the clip is part of its step; any other optimizer gets
    ``clip_grad_norm_`` here as in the reference."""
    from .optim import FusedAdamW
    model.train()
    total, n = 0.0, 0
    for eeg, fmri, labels, _ in loader:
        optimizer.zero_grad()
        loss = criterion(model(eeg.to(device), fmri.to(device)), labels.to(device))
        loss.backward()
        if isinstance(optimizer, FusedAdamW):
            optimizer.max_grad_norm = float(grad_clip)
        elif grad_clip > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        optimizer.step()
        total += loss.item()
        n += 1
    return total / max(n, 1)


class ImprovedTriModalFusionNet(nn.Module):
    """checkpoint wrapper of the bridge pipeline (_test_bridge.py:118-151): state_dict keys carry the
    ``model.`` prefix of ``best_trimodal_fold*.pt``; ``return_feats`` -> dict(logits, gates, fused_feats)."""

    def __init__(self, in_pw_dim, in_erp_dim, in_conn_dim, fusion_dim=128, num_classes=2, dropout=0.3,
                 num_transformer_layers=2, num_heads=4):
        super().__init__()
        from .crossmodal_v4_enhancements import EnhancedTriModalFusionNetV4
        self.model = EnhancedTriModalFusionNetV4(erp_channels=in_erp_dim, pw_channels=in_pw_dim,
                                                 conn_features=in_conn_dim, hidden_dim=fusion_dim,
                                                 num_classes=num_classes, dropout=dropout,
                                                 num_transformer_layers=num_transformer_layers, num_heads=num_heads)
        self.fusion_weight_history = []

    def forward(self, erp, pw, conn, return_feats=False):
        if return_feats:
            logits, gates, fused = self.model(erp, pw, conn, return_fusion_weights=True, return_fused_feats=True)
            return {"logits": logits, "gates": gates, "fused_feats": fused}
        return self.model(erp, pw, conn)

    def get_fusion_weights(self):
        from .crossmodal_v4_enhancements import get_fusion_weights_from_model
        return get_fusion_weights_from_model(self.model)

    def track_fusion_weights(self):
        w = self.get_fusion_weights()
        if w:
            self.fusion_weight_history.append(w)

    def get_weight_history(self):
        return self.fusion_weight_history


@torch.no_grad()
def extract_eeg_features(model, raw_dataset, device, batch_size: int = 64):
    """{subject: mean fused feature over the subject's EEG samples} from a frozen tri-modal model.

    Same contract as the reference's extract_eeg_features (_test_bridge.py:560-585) - items are
    ``(eeg_samples, _, _, label, subject)`` with ``eeg_samples`` a list of (erp, pw, conn) arrays - but
    the samples of ALL subjects run through the GPU in batches of ``batch_size`` instead of one
    forward per sample; the per-subject mean is a segment mean over the batch outputs."""
    model.eval()
    erps, pws, conns, owner, subjects = [], [], [], [], []
    for idx in range(len(raw_dataset)):
        eeg_samples, _, _, _, subj = raw_dataset[idx]
        if not len(eeg_samples):
            continue
        subjects.append(subj)
        for erp_np, pw_np, conn_np in eeg_samples:
            erps.append(torch.as_tensor(erp_np, dtype=torch.float32))
            pws.append(torch.as_tensor(pw_np, dtype=torch.float32))
            conns.append(torch.as_tensor(conn_np, dtype=torch.float32).reshape(-1))
            owner.append(len(subjects) - 1)
    if not erps:
        return {}
    feats = []
    for i in range(0, len(erps), batch_size):
        sl = slice(i, i + batch_size)
        out = model(erp=torch.stack(erps[sl]).to(device), pw=torch.stack(pws[sl]).to(device),
                    conn=torch.stack(conns[sl]).to(device), return_feats=True)
        feats.append(out["fused_feats"].float().cpu())
    feats = torch.cat(feats, dim=0)
    own = torch.tensor(owner)
    sums = torch.zeros(len(subjects), feats.shape[1]).index_add_(0, own, feats)
    counts = torch.bincount(own, minlength=len(subjects)).clamp_min(1).unsqueeze(1)
    means = sums / counts
    return {s: means[i] for i, s in enumerate(subjects)}


@torch.no_grad()
def extract_fmri_features(model, fmri_act, fmri_conn, subject_list, device, batch_size: int = 256):
    """{subject: fused feature} from a frozen fMRIFusionNet (_test_bridge.py:588-603), batched."""
    model.eval()
    subs = [s for s in subject_list if s in fmri_act and s in fmri_conn]
    features = {}
    for i in range(0, len(subs), batch_size):
        chunk = subs[i:i + batch_size]
        act = torch.stack([fmri_act[s].float() for s in chunk]).to(device)
        conn = torch.stack([fmri_conn[s].float() for s in chunk]).to(device)
        _, fused = model(act, conn, return_features=True)
        fused = fused.float().cpu()
        features.update({s: fused[j] for j, s in enumerate(chunk)})
    return features


def _one_hot_of(logits, target_class):
    if target_class is None:
        target_class = logits.argmax(dim=1)
    return torch.zeros_like(logits).scatter_(1, target_class.view(-1, 1), 1.0), target_class


class BridgeGradientSaliency:
    """|d logit_target / d input| for both modalities (bridge_utils.py:158-183 of the reference):
    eval-mode forward and backward run in the HIP kernels (ops.bridge_forward's differentiable form)."""

    def __init__(self, model, device):
        self.model, self.device = model, device

    def compute(self, eeg_feats, fmri_feats, target_class=None):
        self.model.eval()
        eeg = eeg_feats.clone().detach().to(self.device).requires_grad_(True)
        fmri = fmri_feats.clone().detach().to(self.device).requires_grad_(True)
        logits = self.model(eeg, fmri)
        one_hot, _ = _one_hot_of(logits.detach(), target_class if target_class is None else target_class.to(self.device))
        self.model.zero_grad()
        logits.backward(gradient=one_hot)
        return {"eeg": eeg.grad.abs().cpu().numpy(), "fmri": fmri.grad.abs().cpu().numpy()}


class BridgeIntegratedGradients:
    """Integrated gradients from the zero baseline (reference :189-229).  The reference runs its
    ``n_steps`` interpolation points one forward/backward at a time; here they are ONE batch of
    ``n_steps * B`` rows through the same kernels (the bridge has no batch statistics, rows are
    independent), i.e. one launch sequence instead of fifty."""

    def __init__(self, model, device, n_steps=50):
        self.model, self.device, self.n_steps = model, device, n_steps

    def compute(self, eeg_feats, fmri_feats, target_class=None):
        self.model.eval()
        eeg = eeg_feats.detach().to(self.device).float()
        fmri = fmri_feats.detach().to(self.device).float()
        B = eeg.shape[0]
        if target_class is None:                      # the reference fixes the class at alpha = 0 (its first step)
            with torch.no_grad():
                target_class = self.model(torch.zeros_like(eeg), torch.zeros_like(fmri)).argmax(dim=1)
        target_class = target_class.to(self.device)
        alphas = torch.linspace(0.0, 1.0, self.n_steps, device=self.device).view(-1, 1, 1)
        e = (alphas * eeg.unsqueeze(0)).reshape(self.n_steps * B, -1).requires_grad_(True)
        f = (alphas * fmri.unsqueeze(0)).reshape(self.n_steps * B, -1).requires_grad_(True)
        logits = self.model(e, f)
        one_hot, _ = _one_hot_of(logits.detach(), target_class.repeat(self.n_steps))
        self.model.zero_grad()
        logits.backward(gradient=one_hot)
        ge = e.grad.view(self.n_steps, B, -1).mean(dim=0)
        gf = f.grad.view(self.n_steps, B, -1).mean(dim=0)
        return {"eeg": (eeg * ge).abs().cpu().numpy(), "fmri": (fmri * gf).abs().cpu().numpy()}


def collate_bridge(batch):
    """(``_test_bridge.py:755-760``) stack features, long labels, subject list."""
    eeg = torch.stack([b[0] for b in batch])
    fmri = torch.stack([b[1] for b in batch])
    labels = torch.tensor([b[2] for b in batch], dtype=torch.long)
    return eeg, fmri, labels, [b[3] for b in batch]


@torch.no_grad()
def evaluate_bridge(model, loader, device):
    """(``_test_bridge.py:791-820``) -> (metrics, targets, probs, subjects)."""
    import numpy as np
    from .fmri_utils import classification_metrics
    model.eval()
    preds, targets, probs, subjects = [], [], [], []
    for eeg, fmri, labels, subj in loader:
        logits = model(eeg.to(device), fmri.to(device)).float()
        probs.append(torch.softmax(logits, dim=1).cpu().numpy())
        preds.append(logits.argmax(dim=1).cpu().numpy())
        targets.append(labels.numpy())
        subjects.extend(subj)
    preds, targets, probs = np.concatenate(preds), np.concatenate(targets), np.concatenate(probs)
    return classification_metrics(targets, preds, probs, 2), targets, probs, subjects


def balanced_class_weights(labels):
    """sklearn's ``compute_class_weight('balanced')``: n / (n_classes * count_c) for the classes present."""
    import numpy as np
    labels = np.asarray(labels)
    classes, counts = np.unique(labels, return_counts=True)
    return torch.tensor(len(labels) / (len(classes) * counts), dtype=torch.float32)


def run_bridge_loocv(dataset, eeg_dim=128, fmri_dim=64, bridge_dim=128, num_classes=2, dropout=0.3,
                     lr=1e-4, weight_decay=1e-4, batch_size=8, num_epochs=100, patience=15, grad_clip=1.0,
                     device=None, xai=True, ig_steps=50, seed=None):
    """Leave-one-subject-out protocol of the bridge pipeline (``_test_bridge.py:826-970``): per fold a
    fresh ``EEGfMRIBridgeFusionNet``, balanced class weights from the training labels, AdamW +
    ReduceLROnPlateau(0.5, 5) on the epoch's training loss, best-training-loss state restored after
    ``patience`` epochs without improvement, then the held-out subject is scored and (``xai``) explained
    with gradient saliency and integrated gradients - all forward/backward passes on the HIP kernels.

    Returns a dict: ``predictions`` [(subject, true, pred, prob_class1)], ``fusion_weights`` [per-fold
    dict], ``fused_features`` {subject: (bridge_dim,)}, ``saliency`` / ``integrated_gradients``
    {subject: {'eeg','fmri'}}, ``attn_fusion`` {subject: {...}}, ``metrics``."""
    import copy
    import numpy as np
    from torch.utils.data import DataLoader, Subset
    from .crossmodal_eeg_scr import _PlateauLR
    from .fmri_utils import classification_metrics
    from .optim import FusedAdamW
    device = device or torch.device("cuda")
    labels = np.array([s["label"] for s in dataset.samples])
    n = len(dataset)
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    res = {"predictions": [], "fusion_weights": [], "fused_features": {}, "saliency": {},
           "integrated_gradients": {}, "attn_fusion": {}}
    for held in range(n):
        train_idx = [i for i in range(n) if i != held]
        subject = dataset.samples[held]["subject"]
        loader = DataLoader(Subset(dataset, train_idx), batch_size=batch_size, shuffle=True,
                            collate_fn=collate_bridge, generator=gen)
        if seed is not None:
            torch.manual_seed(seed + held)
        model = EEGfMRIBridgeFusionNet(eeg_dim, fmri_dim, bridge_dim, num_classes, dropout=dropout).to(device)
        criterion = WeightedCrossEntropy(balanced_class_weights(labels[train_idx])).to(device)
        opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
        sched = _PlateauLR(opt, factor=0.5, patience=5)
        best, best_state, bad = float("inf"), None, 0
        for _ in range(num_epochs):
            loss = train_bridge_epoch(model, loader, opt, criterion, device, grad_clip)
            sched.step(loss)
            if loss < best:
                best, best_state, bad = loss, copy.deepcopy(model.state_dict()), 0
            else:
                bad += 1
            if bad >= patience:
                break
        if best_state:
            model.load_state_dict(best_state)
            ops.weights_changed()
        model.eval()
        eeg_t, fmri_t, label_t, _ = collate_bridge([dataset[held]])
        with torch.no_grad():
            logits, fused, fw, aw = model(eeg_t.to(device), fmri_t.to(device), return_features=True, return_weights=True)
        probs = torch.softmax(logits.float(), dim=1)
        res["predictions"].append((subject, int(label_t[0]), int(logits.argmax(dim=1)), float(probs[0, 1])))
        res["fused_features"][subject] = fused.squeeze(0).float().cpu()
        res["fusion_weights"].append(model.get_fusion_weights())
        res["attn_fusion"][subject] = {"label": int(label_t[0]), "prediction": int(logits.argmax(dim=1)),
                                       "fusion_weights": fw.squeeze(0).float().cpu().numpy(),
                                       "attn_weights": aw.squeeze(0).float().cpu().numpy()}
        if xai:
            res["saliency"][subject] = {k: v[0] for k, v in BridgeGradientSaliency(model, device).compute(eeg_t, fmri_t).items()}
            res["integrated_gradients"][subject] = {
                k: v[0] for k, v in BridgeIntegratedGradients(model, device, ig_steps).compute(eeg_t, fmri_t).items()}
    t = np.array([p[1] for p in res["predictions"]])
    pr = np.array([p[2] for p in res["predictions"]])
    p1 = np.array([p[3] for p in res["predictions"]])
    res["metrics"] = classification_metrics(t, pr, np.stack([1 - p1, p1], axis=1), 2)
    return res

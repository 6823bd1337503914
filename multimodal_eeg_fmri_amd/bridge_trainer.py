"""Data-parallel contrastive EEG<->fMRI bridge trainer (north-star hot path).

One process per GPU (``torch.distributed`` backend ``nccl`` == RCCL over xGMI).
Per step and per rank: EEG temporal encoder + fMRI voxel encoder forward on the
local (EEG-epoch, fMRI-volume) pairs, projection heads, one all-gather of the
packed L2-normalised embeddings (global contrastive negatives), fused
similarity/InfoNCE forward+backward on this rank's rows of the gathered batch
(every rank evaluates all rows, so no reduce-scatter of column gradients is
issued), encoder backward, the all-reduce of the flat fp32 gradient bucket in
one piece per FINISHED LAYER GROUP (``groups``: the fMRI encoder as soon as its
shorter backward is done; transformer stack + heads, then conv blocks 3-2, as
soon as their handed-over weight-gradient sums have been flushed on the side
stream; only conv block 1 - 0.1 MB - after the chain), each an asynchronous
collective the optimizer alone waits for, and the fused clip+AdamW on the flat
parameter bucket.  At world size > 1 the whole step, collectives included, is
ONE hipGraph when the backend's collectives can be captured (RCCL) - all ranks
agree on that outcome before any replay (``dp.agree_on_capture``) - else three
graph segments around two eager collectives (``capture_mode`` says which).
BatchNorm statistics stay per-rank (the reference has no SyncBN; SURVEY.md
section 8e).

The reference has no such trainer (its loops are run_training_lite.py:474-489
and _test_bridge.py:775-788: zero_grad / forward / backward / clip 1.0 / AdamW);
this keeps that step structure.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _hip, dp, ops
from .autograd import GradBag, contrastive_embed_bwd, deferred, erp_encoder_bwd, power_encoder_bwd, volume_encoder_bwd
from .bridge_utils import EEGfMRIContrastiveBridge
from .enhanced_models_v4 import EnhancedERPEncoder
from .fmri_utils import fMRIVolumeEncoder3D
from .optim import FlatBucket


class BridgeTrainer(nn.Module):
    def __init__(self, eeg_channels: int = 64, hidden_dim: int = 128, fmri_dim: int = 64,
                 bridge_dim: int = 128, dropout: float = 0.3, lr: float = 1e-4,
                 weight_decay: float = 1e-4, grad_clip: float = 1.0, betas=(0.9, 0.999),
                 eps: float = 1e-8, group=None, device="cuda", mode: str = "graph", eeg_encoder: Optional[nn.Module] = None):
        """``eeg_encoder``: the EEG branch when it is not the default ``EnhancedERPEncoder(eeg_channels, hidden_dim, 2, 4,
        dropout)`` - an ``EnhancedPowerEncoder`` (enhanced_models_v4.py:196-285) or a ``MultiScaleSTFTPowerEncoder``
        (BASELINE config #5: raw EEG -> multi-scale STFT power -> a4); it must end in ``hidden_dim`` features."""
        super().__init__()
        self.eeg_encoder = EnhancedERPEncoder(eeg_channels, hidden_dim, 2, 4, dropout) if eeg_encoder is None else eeg_encoder
        from .crossmodal_v4_enhancements import MultiScaleSTFTPowerEncoder
        from .enhanced_models_v4 import EnhancedPowerEncoder
        if isinstance(self.eeg_encoder, EnhancedERPEncoder):
            self._eeg_kind = "erp"
        elif isinstance(self.eeg_encoder, MultiScaleSTFTPowerEncoder):
            self._eeg_kind = "stft"
        elif isinstance(self.eeg_encoder, EnhancedPowerEncoder):
            self._eeg_kind = "power"
        else:
            raise TypeError(f"BridgeTrainer: no tape for an EEG encoder of type {type(self.eeg_encoder).__name__}")
        self.fmri_encoder = fMRIVolumeEncoder3D(1, fmri_dim, dropout=dropout)
        self.head = EEGfMRIContrastiveBridge(hidden_dim, fmri_dim, bridge_dim, dropout)
        self.to(device)
        self.group = group
        self.two_streams = True
        self.mode = mode
        self._cap = None
        self._weight_list = None                      # recorded by the first manual step
        self._arena_need = None                       # floats of scratch one step uses (None: clear all)
        self.stamps = None                            # int64[16] device buffer when phase stamps are wanted
        self.force_segments = False                   # rehearsal: run the N > 1 segmented step even at world 1
        import os
        # MM_ONE_STREAM=1 (diagnostic): run the fMRI branch on the main stream after the EEG branch
        self._one_stream = bool(os.environ.get("MM_ONE_STREAM"))
        self._side_stream = None                      # created on first use (the host logic also constructs on CPU)
        self.lr, self.weight_decay, self.grad_clip = lr, weight_decay, grad_clip
        self.betas, self.eps = betas, eps
        br = self.head.bridge
        # only what the contrastive path touches is trained (the classifier /
        # cross-attention half of the bridge stays out of the bucket)
        for name, p in br.named_parameters():
            if not (name.startswith("eeg_proj") or name.startswith("fmri_proj")):
                p.requires_grad_(False)
        # bucket order = the order in which gradients are NOT yet final, i.e. the reverse of when each layer group's
        # all-reduce can start (`_seg_backward`): conv block 1 (last kernel of the chain) | conv blocks 2-3 (final once
        # the second hand-over has been flushed) | transformer stack + encoder head + projection heads + logit scale
        # (first hand-over) | fMRI encoder (its own, shorter backward).  Each group is one contiguous range of the
        # flat bucket, the fMRI encoder's the tail [fmri_lo, n).
        head_params = [p for p in br.parameters() if p.requires_grad] + [self.head.logit_scale]
        if self._eeg_kind == "erp":
            cl = self.eeg_encoder.conv_layers
            conv1 = list(cl[0].parameters()) + list(cl[1].parameters())
            conv23 = [p for i in (4, 5, 9, 10) for p in cl[i].parameters()]
            seen = {id(p) for p in conv1 + conv23}
            rest = [p for p in self.eeg_encoder.parameters() if id(p) not in seen]
            layout = [("eeg conv block 1", "main", conv1), ("eeg conv blocks 2-3", "handed1", conv23),
                      ("eeg transformer stack + heads", "handed0", rest + head_params)]
        else:
            layout = [("eeg encoder + heads", "main", list(self.eeg_encoder.parameters()) + head_params)]
        layout.append(("fmri encoder", "fmri", list(self.fmri_encoder.parameters())))
        # MM_DP_GROUPS=2 (A/B knob): the round-3 split - the fMRI slice early, everything else after the chain
        if os.environ.get("MM_DP_GROUPS", "4") == "2" and len(layout) > 2:
            layout = [("eeg encoder + heads", "main", [p for _, _, ps in layout[:-1] for p in ps]), layout[-1]]
        self.bucket = FlatBucket([p for _, _, ps in layout for p in ps])
        self.groups = []                              # (name, ready point, lo, hi) over the flat bucket, in bucket order
        off = 0
        for name, ready, ps in layout:
            k = sum(p.numel() for p in ps if p.requires_grad)
            self.groups.append((name, ready, off, off + k))
            off += k
        assert off == self.bucket.n
        self.fmri_lo = self.groups[-1][2]
        self._works = []                              # asynchronous all-reduces of the running step (waited for by the optimizer)
        self.capture_mode = None                      # "one graph" | "one graph + captured RCCL collectives" | "3 segments + 2 eager collectives"
        self.bucket.state[2] = lr
        # {loss, top-1 e->f, top-1 f->e, d loss / d logit_scale} of the last step: owned by this trainer
        # (plain stores of the loss kernel), so the tensors a step returns are overwritten only by THIS
        # trainer's next step - keep a value with .item() / .clone()
        self._scal = torch.zeros(4, device=device)
        ops.weights_changed()

    @property
    def _side(self):
        if self._one_stream:
            return torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        return self._side_stream

    @property
    def world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if self.group is not None else 1

    def set_lr(self, lr: float):
        self.lr = lr
        self.bucket.state[2] = lr

    def forward(self, eeg: torch.Tensor, fmri: torch.Tensor):
        """the two encoders are independent until the heads: they run on two HIP
        streams (autograd replays each backward on its forward stream), so the
        many sub-chip kernels of one branch overlap the other's latency."""
        if not self.two_streams:
            return self.head(self.eeg_encoder(eeg), self.fmri_encoder(fmri), self.group)
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            ff = self.fmri_encoder(fmri)
        fe = self.eeg_encoder(eeg)
        main.wait_stream(self._side)
        ff.record_stream(main)
        return self.head(fe, ff, self.group)

    # ------------------------------------------------------------------ step
    def train_step(self, eeg: torch.Tensor, fmri: torch.Tensor) -> Dict[str, torch.Tensor]:
        """zero_grad -> forward -> backward -> (all-reduce) -> clip + AdamW.

        ``mode``: "graph" (default) replays the step from hipGraphs captured on
        first use (one graph at world 1; three segments around the two
        collectives otherwise); "manual" runs the same autograd-free tape eagerly;
        "autograd" goes through the public nn.Module / torch.autograd surface."""
        if self.mode == "autograd":
            return self._step_autograd(eeg, fmri)
        if self.mode == "manual":
            with torch.no_grad():
                return self._step_manual(eeg, fmri)
        return self._step_graph(eeg, fmri)

    def _step_autograd(self, eeg, fmri):
        b = self.bucket
        b.zero_grad()
        loss, acc_e, acc_f = self.forward(eeg, fmri)
        loss.backward()
        b.absorb_autograd_grads()
        self._seg_optimizer()
        return {"loss": loss.detach(), "top1_e2f": acc_e, "top1_f2e": acc_f}

    # ---- the segments of the autograd-free tape ------------------------------
    STAMP_NAMES = ("step start", "weights prepared", "EEG fwd done", "fMRI fwd start", "fMRI fwd done",
                   "heads fwd done", "loss done", "heads bwd done", "EEG bwd done", "fMRI bwd start",
                   "fMRI bwd done", "grad reductions done", "AdamW done", "side stream done")

    def _stamp(self, i):
        if self.stamps is not None:
            _hip.call("mm_debug_stamp", self.stamps, i)

    def _seg_forward(self, eeg, fmri, xb=None):
        """``xb``: the EEG batch already packed (B, T, Cp) bf16 (the captured step reads its static packed buffer, which
        the trainer fills outside the graph together with the input copies)"""
        self._stamp(0)
        # one memset of what the step's accumulators actually use (high-water mark of the first
        # step + slack); the gradient bucket is cleared by the previous step's AdamW kernel
        if self._weight_list is not None:
            # every bf16 weight image of the step AND the clearing of its accumulators: one launch
            ops.arena.begin(eeg.device, clear=self._arena_need, defer_zero=True)
            ops.weights.prepare_all(self._weight_list, zero=ops.arena.zero_range())
        else:
            ops.arena.begin(eeg.device, clear=self._arena_need)
        self._stamp(1)
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)                 # fork point: recorded before any encoder kernel
        # The longer branch is issued FIRST: a hipGraph replay writes its kernel packets in capture order at
        # ~4.6 us per node, and the branch captured second cannot start before the host has written every packet
        # of the first (profiles/README.md).  That is the EEG chain at 32^3 voxels, the fMRI branch at config #4's
        # 64 x 64 x 48 (`_fmri_is_longer`).
        self._fmri_longer = self._fmri_is_longer(fmri)

        def fmri_branch():
            with torch.cuda.stream(self._side):
                self._stamp(3)
                out = ops._vol_forward_impl(self.fmri_encoder, fmri, True, True)
                self._stamp(4)
            return out
        if self._fmri_longer:
            ff, sv_f = fmri_branch()
        fe, sv_e = self._eeg_forward(eeg, xb)
        self._stamp(2)
        if not self._fmri_longer:
            ff, sv_f = fmri_branch()
        main.wait_stream(self._side)
        z, sv_h = ops.contrastive_embed_impl(self.head.bridge, fe, ff, True)
        self._stamp(5)
        return z, (sv_e, sv_f, sv_h)

    @staticmethod
    def _fmri_is_longer(fmri) -> bool:
        """the voxel branch outlasts the EEG chain (config #4: 64 x 64 x 48 = 6 x the voxels of 32^3, 1.4 ms of kernels
        against 1.1 ms): its stream then takes no work from the chain and is issued first.  MM_FMRI_LONGER=0/1 overrides."""
        env = os.environ.get("MM_FMRI_LONGER")
        if env is not None:
            return env == "1"
        return fmri[0].numel() >= 4 * 32 ** 3

    def _eeg_forward(self, eeg, xb):
        enc = self.eeg_encoder
        if self._eeg_kind == "erp":
            return ops._erp_forward_impl(enc, eeg, True, True, xb=xb)
        if self._eeg_kind == "power":
            return ops._power_forward_impl(enc, xb if xb is not None else ops.pack_nct(eeg), True, False)
        # config #5: the parameter-free multi-scale STFT power front-end (z-scored per sample), then a4
        spec = ops.stft_front_end(eeg, enc.n_ffts, enc.hop, enc.normalize)
        return ops._power_forward_impl(enc.encoder, spec, True, False)

    def _seg_loss(self, z_all, scal, dz):
        """symmetric InfoNCE of this rank's rows against the gathered batch, gradient w.r.t. ITS rows only
        (``mm_clip_loss_own_rows``: every rank evaluates all rows of the gathered batch, so no reduce-scatter
        of column gradients is needed).  ``scal`` = the trainer's own 4-float result buffer, ``dz`` (B, 2N): both written with plain stores."""
        N2 = z_all.shape[1]
        B = dz.shape[0]
        ls = self.head.logit_scale.detach().reshape(1)
        ws = ops._empty((6 * z_all.shape[0],), torch.float32, z_all)
        _hip.call("mm_clip_loss_own_rows", z_all, ls, scal, dz, ws, B, z_all.shape[0], N2 // 2, dp.rank(self.group) * B)
        self._stamp(6)

    def _reduce_group(self, ready: str):
        """all-reduce (asynchronous: the issuing stream does not wait) every bucket range whose gradients are final at
        ``ready``; the handles are waited for by `_seg_optimizer`.  Host issue order = collective order on every rank."""
        for name, rdy, lo, hi in self.groups:
            if rdy == ready and hi > lo:
                self._works.append(dp.allreduce_sum_(self.bucket.g[lo:hi], self.group, async_op=True))

    def _seg_backward(self, saved, dz, scal, reduce: bool = False):
        """``reduce``: all-reduce every layer group of the gradient bucket as soon as it is final - the fMRI encoder's
        after that branch's backward, the transformer stack's and conv blocks 3-2's after their handed-over sums were
        flushed on the side stream (all three hidden beside the EEG chain), conv block 1's after the chain"""
        sv_e, sv_f, sv_h = saved
        self._works = []
        bag = GradBag()
        with deferred(bag, dz.device):           # ONE batched reduction after both branches joined
            # d loss / d logit_scale (scal[3]) rides in the same batched reduction launch
            bag.defer(scal.data_ptr() + 12, self.head.logit_scale._mm_grad.view(1), 1, 1, 1, keep=scal)
            dfe, dff = contrastive_embed_bwd(bag, sv_h, dz)
            self._stamp(7)
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)
            # the transformer stack's weight-gradient slot sums and parameter reductions (~50 MB of
            # reads) do not wait for the end of the chain: they are handed to the side stream, which
            # is idle once the fMRI branch is done - unless that branch is the longer one (`_fmri_is_longer`): then
            # nothing is handed over, the chain flushes its own sums, and the fMRI backward is issued first
            hand = not getattr(self, "_fmri_longer", False)
            handed = []

            def split():
                if not hand:
                    return
                ev = torch.cuda.Event()
                handed.append((bag.hand_over(), ev))
                ev.record()
            # second hand-over: the weight gradients of conv blocks 3 and 2 (k = 3, 5) leave the chain too;
            # block 1's (the last kernel of the chain) stays, and its slot sum runs on this stream BEFORE
            # the join below instead of after it
            def split_convs():
                split()
                bag.defer_conv_wgrads = False
            # MM_CONV_WGRADS_HANDED: 1 = block 2's only (default: block 3's weight gradient stays on the chain), 2 = blocks 3
            # and 2 (round 2 / early round 3, when the chain was the later stream), 0 = none.  Which stream ends later
            # decides: 0.773-0.776 / 0.782-0.799 / 0.789-0.793 ms per step for 1 / 2 / 0 (profiles/r03_second_half_ab.txt)
            handed_convs = int(os.environ.get("MM_CONV_WGRADS_HANDED", "1")) if hand else 0
            bag.defer_conv_wgrads = handed_convs >= 2

            def conv3_done():
                bag.defer_conv_wgrads = handed_convs >= 1

            def fmri_branch():
                with torch.cuda.stream(self._side):
                    self._stamp(9)
                    bag_f = GradBag()                    # the fMRI branch flushes its own reductions on ITS stream,
                    with deferred(bag_f, dz.device):     # hidden beside the rest of the EEG backward
                        volume_encoder_bwd(bag_f, sv_f, dff)
                    self._stamp(10)
                    if reduce and hand:
                        self._reduce_group("fmri")
                    for i, (hb, ev) in enumerate(handed):
                        self._side.wait_event(ev)
                        hb.flush(dz.device)
                        if reduce:
                            self._reduce_group(f"handed{i}")
                    self._stamp(13)
                return bag_f
            if not hand:
                bag_f = fmri_branch()
            finish = None
            if self._eeg_kind == "erp":
                erp_encoder_bwd(bag, sv_e, dfe, after_blocks=split, after_conv2=split_convs, after_conv3=conv3_done)   # longer chain first (see _seg_forward)
            else:
                _, finish = power_encoder_bwd(bag, sv_e, dfe)
            bag.flush(dz.device)
            if finish is not None:
                finish()                         # the merged 192-channel conv / BatchNorm gradients back into the six real parameters
            self._stamp(8)
            if hand:
                bag_f = fmri_branch()
            if reduce:
                if not hand:                     # the chain's groups are final first; the fMRI encoder's goes out last
                    self._reduce_group("handed0")
                    self._reduce_group("handed1")
                self._reduce_group("main")       # issued last on the host: the collectives run in issue order
                if not hand:
                    with torch.cuda.stream(self._side):
                        self._reduce_group("fmri")
            main.wait_stream(self._side)
        self._stamp(11)
        self._bags = getattr(self, "_bags", [])[-12:] + [bag, bag_f] + [hb for hb, _ in handed]   # keep descriptor tables alive

    def _seg_optimizer(self, reduced: bool = False):
        """``reduced``: the backward issued the per-group all-reduces (wait for them); else ONE all-reduce of the whole bucket"""
        if reduced:
            for w in self._works:
                dp.wait(w)
            self._works = []
        else:
            dp.allreduce_sum_(self.bucket.g, self.group)
        self._seg_adamw()

    def _seg_adamw(self):
        b = self.bucket
        _hip.call("mm_sumsq", b.g, b.state, b.n)
        _hip.call("mm_adamw_clip", b.p, b.g, b.m, b.v, b.state, b.n, self.betas[0], self.betas[1],
                  self.eps, self.weight_decay, self.grad_clip, 1.0 / self.world, 1, ops.EP())
        ops.weights_changed()
        if self._arena_need is None:
            self._arena_need = (ops.arena.high * 5 // 4 + 4095) // 4096 * 4096
        ops.arena.end()
        self._stamp(12)

    def _step_manual(self, eeg, fmri):
        recording = self._weight_list is None
        if recording:                                 # first step: note every weight image the tape asks for
            ops.weights.start_recording()
        try:
            return self._step_manual_body(eeg, fmri)
        finally:
            if recording:
                self._weight_list = ops.weights.stop_recording()

    def _step_manual_body(self, eeg, fmri):
        z, saved = self._seg_forward(eeg, fmri)
        z_all = dp.gather_embeddings(z, self.group)
        scal, dz = self._scal, ops._empty(tuple(z.shape), torch.float32, z)
        self._seg_loss(z_all, scal, dz)
        early = dp.active(self.group)
        self._seg_backward(saved, dz, scal, reduce=early)
        self._seg_optimizer(reduced=early)
        return {"loss": scal[0], "top1_e2f": scal[1], "top1_f2e": scal[2]}

    # ---- hipGraph capture ------------------------------------------------------
    def _capture(self, eeg, fmri):
        dev = eeg.device
        world = self.world
        c = {"epoch": torch.zeros(1, dtype=torch.int32, device=dev)}
        # the step's static inputs are two views of ONE flat buffer `c["in"]` = [EEG operand | fMRI volumes fp32], so that a
        # host-fed loop fills them with a single copy (`train_step_packed`).  EEG operand: the first convolution's packed
        # bf16 (B, T, Cp) image (the STFT front-end reads the raw fp32 batch instead and keeps that)
        stft = self._eeg_kind == "stft"
        Bx, Cx, Tx = eeg.shape
        n_e = eeg.numel() * 4 if stft else Bx * Tx * ops.cpad(Cx) * 2
        c["in"] = torch.empty(n_e + fmri.numel() * 4, dtype=torch.uint8, device=dev)
        c["fmri"] = c["in"][n_e:].view(torch.float32).view(fmri.shape)
        c["fmri"].copy_(fmri)
        if stft:
            c["eeg"], c["xb"] = c["in"][:n_e].view(torch.float32).view(eeg.shape), None
            c["eeg"].copy_(eeg)
        else:
            c["eeg"] = eeg.clone()                      # (shape carrier / in-place loader target; the graph reads c["xb"])
            c["xb"] = c["in"][:n_e].view(torch.bfloat16).view(Bx, Tx, ops.cpad(Cx))
            _hip.call("mm_pack_nct_bf16", c["eeg"], c["xb"], Bx, Cx, Tx, c["xb"].shape[2])
        ops.set_seed_epoch(c["epoch"])
        # warm-up outside capture (lazy inits, allocator priming) on a snapshot of the
        # training state, so that the first replay is really step 1
        b = self.bucket
        snap = [t.clone() for t in (b.p, b.m, b.v, b.state)]
        bufs = [t for t in self.buffers() if t is not None]
        snap_bufs = [t.clone() for t in bufs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                self._step_manual(c["eeg"], c["fmri"])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for t, sv in zip((b.p, b.m, b.v, b.state), snap):
                t.copy_(sv)
            for t, sv in zip(bufs, snap_bufs):
                t.copy_(sv)
        ops.weights_changed()                                     # weight-image kernels must be recorded
        c["pool"] = torch.cuda.graph_pool_handle()
        graphs = []

        # A process group's watchdog thread polls the events of its finished collectives (hipEventQuery) for a while after
        # they are done; under a GLOBAL-mode capture that call is refused and the watchdog ends the process ("operation
        # not permitted when stream is capturing" - seen once in the aborted-capture rehearsal, where the communicator
        # check runs right before the segments are recorded).  Every capture of a distributed job is therefore
        # thread-local: only this thread's calls are policed.
        dist_alive = torch.distributed.is_available() and torch.distributed.is_initialized()

        def record(fn, mode=None):
            mode = mode or ("thread_local" if dist_alive else "global")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=c["pool"], capture_error_mode=mode), torch.no_grad():
                fn()
            graphs.append(g)

        B = eeg.shape[0]
        N2 = 2 * self.head.bridge.bridge_dim
        c["scal"] = self._scal
        import os
        dist_step = not (world == 1 and not (self.force_segments and self.group is not None))
        if not dist_step:
            def whole():
                z, saved = self._seg_forward(c["eeg"], c["fmri"], xb=c["xb"])
                c["z"] = z
                c["dz"] = ops._empty(tuple(z.shape), torch.float32, z)
                self._seg_loss(z, c["scal"], c["dz"])
                self._seg_backward(saved, c["dz"], c["scal"])
                self._grad_probe()
                self._seg_adamw()
            record(whole)
            self.capture_mode = "one graph"
        elif dp.CAPTURABLE and os.environ.get("MM_DP_CAPTURE", "1") != "0" and self._capture_with_collectives(c, record, graphs, B, N2, world, dev):
            self.capture_mode = "one graph + captured RCCL collectives"
        else:
            self.capture_mode = "3 segments + 2 eager collectives"
            def seg1():
                c["z"], c["saved"] = self._seg_forward(c["eeg"], c["fmri"], xb=c["xb"])
                c["dz"] = ops._empty((B, N2), torch.float32, c["z"])
            record(seg1)
            c["z_all"] = torch.empty(world * B, N2, device=dev)

            def seg2():
                self._seg_loss(c["z_all"], c["scal"], c["dz"])
                self._seg_backward(c["saved"], c["dz"], c["scal"])
            record(seg2)

            def seg3():
                self._grad_probe()
                self._seg_adamw()
            record(seg3)
        c["graphs"] = graphs
        self._cap = c

    def _grad_probe(self):
        """``self.grad_probe`` (a flat fp32 tensor the size of the bucket, set by a test BEFORE the first graph step):
        the step copies its finished gradients there just before clip + AdamW clears them - one more node, the
        arithmetic untouched - so that a replayed step's gradients can be compared with the oracle's"""
        probe = getattr(self, "grad_probe", None)
        if probe is not None:
            probe.copy_(self.bucket.g)

    def _capture_with_collectives(self, c, record, graphs, B, N2, world, dev) -> bool:
        """the N > 1 step as ONE hipGraph: the all-gather of the embeddings and the all-reduce of every layer group of the
        gradient bucket (side branch / chain, `_seg_backward`) are graph nodes (no replay gaps, no host in the step).
        Every rank then reports its outcome and ALL take the same form (`dp.agree_on_capture`): the graph when every rank
        captured it, else the three segments on every rank (returns False, nothing recorded).  A recorded collective has
        not run, so dropping the graphs leaves the ranks in step; when the abort came after collectives had been recorded,
        or only on some ranks, the communicator must first pass `dp.check_communicator`."""
        c["z_all"] = torch.empty(world * B, N2, device=dev)
        # the communicator must exist before the capture starts (its lazy initialisation is not capturable)
        dp.all_gather_into(c["z_all"], torch.zeros(B, N2, device=dev), self.group)
        dp.wait(dp.allreduce_sum_(torch.zeros(8, device=dev), self.group, async_op=True))
        torch.cuda.synchronize()

        def whole_dp():
            z, saved = self._seg_forward(c["eeg"], c["fmri"], xb=c["xb"])
            c["z"] = z
            c["dz"] = ops._empty((B, N2), torch.float32, z)
            dp.all_gather_into(c["z_all"], z, self.group)
            self._seg_loss(c["z_all"], c["scal"], c["dz"])
            self._seg_backward(saved, c["dz"], c["scal"], reduce=True)
            for w in self._works:
                dp.wait(w)
            self._works = []
            self._grad_probe()
            self._seg_adamw()
        issued0 = dp.issued
        err = None
        try:
            # thread-local capture mode: the process group's watchdog thread polls its events while this thread captures
            record(whole_dp, mode="thread_local")
        except Exception as e:  # noqa: BLE001 - any refusal (RCCL, the caching allocator, a host sync)
            # keep the text only: the traceback holds `record`'s frame and with it the half-built graph object, and a
            # graph destroyed by the cyclic collector DURING the next capture ends the process (~CUDAGraph: "operation
            # not permitted when stream is capturing")
            err = f"{type(e).__name__}: {e}"
            e.__traceback__ = None
            del e
        verdict = dp.agree_on_capture(err is None, dp.issued - issued0, self.group)
        if verdict == "captured":
            return True
        import gc
        import warnings
        warnings.warn(f"collectives not captured into the step's hipGraph on every rank (this rank: "
                      f"{'captured' if err is None else err}; group verdict {verdict}); "
                      "all ranks use three graph segments around two eager collectives")
        if err is None:
            graphs.pop()                                          # this rank's graph is dropped with the others'
        c.pop("z", None), c.pop("dz", None)
        gc.collect()                                              # every graph of the attempt is gone before the segments are recorded
        torch.cuda.synchronize()
        c["pool"] = torch.cuda.graph_pool_handle()                # (the attempt's memory pool went with its last graph)
        self._works = []
        if verdict == "segments-after-abort":
            dp.check_communicator(self.group, dev)
        ops.arena.end()
        ops.weights_changed()
        return False

    def _step_graph(self, eeg, fmri):
        if self._cap is None or self._cap["eeg"].shape != eeg.shape or self._cap["fmri"].shape != fmri.shape:
            self._capture(eeg, fmri)
        c = self._cap
        ce, cf = eeg.data_ptr() != c["eeg"].data_ptr(), fmri.data_ptr() != c["fmri"].data_ptr()
        if (c["xb"] is not None and ce and cf and eeg.dtype == torch.float32 and fmri.dtype == torch.float32 and eeg.is_cuda and fmri.is_cuda
                and eeg.is_contiguous() and fmri.is_contiguous() and eeg.numel() % 4 == 0 and fmri.numel() % 4 == 0
                and (eeg.data_ptr() | fmri.data_ptr()) % 16 == 0):
            # ONE launch: EEG batch packed into the first convolution's bf16 operand, fMRI batch copied
            Bx, Cx, Tx = eeg.shape
            # (no fp32 copy of the EEG batch: the captured step reads the packed operand only - c["eeg"] gives it its shape)
            _hip.call("mm_stage_inputs", eeg, c["xb"], None, Bx, Cx, Tx, c["xb"].shape[2], c["fmri"], fmri, fmri.numel())
        else:                                               # each input on its own: a loader may fill only one in place
            if ce:
                c["eeg"].copy_(eeg)
            if cf:
                c["fmri"].copy_(fmri)
            if c["xb"] is not None:
                Bx, Cx, Tx = c["eeg"].shape
                _hip.call("mm_pack_nct_bf16", c["eeg"], c["xb"], Bx, Cx, Tx, c["xb"].shape[2])
        return self._replay()

    def _replay(self):
        c = self._cap
        g = c["graphs"]
        if len(g) == 1:
            g[0].replay()
        else:
            g[0].replay()                                              # forward
            dp.all_gather_into(c["z_all"], c["z"], self.group)
            g[1].replay()                                              # loss on the gathered batch + backward
            dp.allreduce_sum_(self.bucket.g, self.group)
            g[2].replay()                                              # clip + AdamW
        return {"loss": c["scal"][0], "top1_e2f": c["scal"][1], "top1_f2e": c["scal"][2]}

    # ---- host-fed input path ---------------------------------------------------
    def pack_host_batch(self, eeg: torch.Tensor, fmri: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """CPU side of the host-fed loop (a data loader's job, off the step's critical path): one (EEG, fMRI) batch as
        ONE flat pinned byte buffer in the layout of the step's static inputs - the EEG epochs already in the first
        convolution's operand format (channels-last (B, T, Cp) bf16, zero-padded channels: half the bytes of fp32, and
        no pack launch on the device; raw fp32 for the STFT front-end) followed by the fp32 volumes.  Round-to-nearest-
        even on the host = what mm_pack_nct_bf16 does on the device: the step's arithmetic is bit-identical.
        Reference counterpart: the ``.to(device)`` of each batch, run_training_lite.py:480-481."""
        eeg, fmri = eeg.detach().cpu().float(), fmri.detach().cpu().float().contiguous()
        if self._eeg_kind == "stft":
            e_bytes = eeg.contiguous().view(-1).view(torch.uint8)
        else:
            B, C, T = eeg.shape
            cp = ops.cpad(C)
            xb = torch.zeros(B, T, cp, dtype=torch.bfloat16)
            xb[:, :, :C] = eeg.permute(0, 2, 1)
            e_bytes = xb.view(-1).view(torch.uint8)
        f_bytes = fmri.view(-1).view(torch.uint8)
        n = e_bytes.numel() + f_bytes.numel()
        if out is None:
            out = torch.empty(n, dtype=torch.uint8).pin_memory()
        out[:e_bytes.numel()].copy_(e_bytes)
        out[e_bytes.numel():].copy_(f_bytes)
        return out

    def train_step_packed(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """one graph-replayed step on a batch that is already in the packed layout of `pack_host_batch` and on the device
        (a staging buffer an H2D copy filled): ONE device-to-device copy into the static inputs, then the replay"""
        if self._cap is None:
            raise RuntimeError("train_step_packed: run one train_step(eeg, fmri) first (it captures the step and fixes the shapes)")
        c = self._cap
        if flat.dtype != torch.uint8 or flat.numel() != c["in"].numel() or not flat.is_cuda:
            raise ValueError(f"train_step_packed: expected a device uint8 buffer of {c['in'].numel()} bytes")
        c["in"].copy_(flat, non_blocking=True)
        return self._replay()

    def host_feeder(self, depth: int = 3) -> "HostFeeder":
        """the loop that feeds `train_step_packed` from pinned host buffers (see `HostFeeder`)"""
        return HostFeeder(self, depth)

    def time_collectives(self, batch: int, iters: int = 50) -> Dict[str, float]:
        """microseconds per call of every collective one step issues, each alone at its message size (HIP events on the
        current stream around ``iters`` back-to-back calls; collective: every rank of the group must call it).  What the
        step pays is less: all but the all-gather and the last group run beside the backward."""
        if not dp.active(self.group):
            return {}
        dev = self.bucket.g.device
        N2 = 2 * self.head.bridge.bridge_dim
        z = torch.zeros(batch, N2, device=dev)
        z_all = torch.empty(self.world * batch, N2, device=dev)
        cases = [(f"all_gather embeddings ({batch}x{N2} fp32 per rank)", lambda: dp.all_gather_into(z_all, z, self.group))]
        for name, _, lo, hi in self.groups:
            buf = torch.zeros(hi - lo, device=dev)
            cases.append((f"all_reduce {name} ({(hi - lo) * 4 / 1e6:.2f} MB)",
                          lambda buf=buf: dp.wait(dp.allreduce_sum_(buf, self.group, async_op=True))))
        out = {}
        for name, fn in cases:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            out[name] = a.elapsed_time(b) / iters * 1e3
        return out

    def input_buffers(self):
        """the static (eeg, fmri) tensors the captured step reads, or None before the first graph step:
        a loader that writes the next batch straight into them saves the two device copies per step"""
        return None if self._cap is None else (self._cap["eeg"], self._cap["fmri"])

    @torch.no_grad()
    def evaluate(self, eeg, fmri):
        was = self.training
        self.eval()
        ops.weights_changed()                      # graph replays bypass the python-side version counter
        try:
            loss, acc_e, acc_f = self.forward(eeg, fmri)
        finally:
            self.train(was)
        return {"loss": loss, "top1_e2f": acc_e, "top1_f2e": acc_f}


class HostFeeder:
    """Feeds `BridgeTrainer.train_step_packed` from packed pinned host buffers (`pack_host_batch`), one H2D copy per
    batch on a copy stream into a ring of ``depth`` staging buffers, overlapped with the steps before it.

    The ordering between the copy stream and the step's stream is kept by the HOST: `step()` blocks on the event its
    batch's copy recorded (issued a whole step earlier), `upload()` on the event recorded after the step that last read
    the staging buffer (``depth - 1`` steps earlier).  Neither queue ever holds a barrier for the other: with
    hipStreamWaitEvent in both directions the same loop lost 8 % to the two cross-queue waits per step, ordered from the
    host it loses 1 % (0.786 vs 0.779 ms, profiles/r04_h2d_probe.txt) - the host has the time, a step costs it 0.25 ms.

        feeder = trainer.host_feeder()
        feeder.upload(packed[0])
        for i in range(n):
            if i + 1 < n:
                feeder.upload(packed[i + 1])      # returns at once; the pinned buffer is free again when the event it returns is done
            out = feeder.step()

    Reference counterpart: the per-batch ``.to(device)`` + step of run_training_lite.py:474-489."""

    def __init__(self, trainer: BridgeTrainer, depth: int = 3):
        if trainer._cap is None:
            raise RuntimeError("HostFeeder: run one train_step(eeg, fmri) first (it captures the step and fixes the shapes)")
        if depth < 2:
            raise ValueError("HostFeeder: at least two staging buffers")
        n = trainer._cap["in"].numel()
        dev = trainer._cap["in"].device
        self.trainer, self.depth, self.nbytes = trainer, depth, n
        self.ring = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.done = [torch.cuda.Event() for _ in range(depth)]
        self.uploaded = 0
        self.stepped = 0

    def upload(self, packed: torch.Tensor) -> "torch.cuda.Event":
        """start the H2D copy of the next batch; ``packed``: a pinned uint8 buffer from `pack_host_batch`.  Returns the
        event that marks the copy's end (until then the pinned buffer must not be rewritten)."""
        if packed.dtype != torch.uint8 or packed.numel() != self.nbytes or packed.is_cuda:
            raise ValueError(f"HostFeeder.upload: expected a host uint8 buffer of {self.nbytes} bytes")
        if self.uploaded - self.stepped >= self.depth:
            raise RuntimeError(f"HostFeeder.upload: {self.depth} batches are already waiting for their step")
        b = self.uploaded % self.depth
        if self.uploaded >= self.depth:
            self.done[b].synchronize()                    # the step that read this staging buffer has finished with it
        with torch.cuda.stream(self.copy_stream):
            self.ring[b].copy_(packed, non_blocking=True)
            self.ready[b].record(self.copy_stream)
        self.uploaded += 1
        return self.ready[b]

    def step(self) -> Dict[str, torch.Tensor]:
        """one training step on the oldest uploaded batch"""
        if self.stepped >= self.uploaded:
            raise RuntimeError("HostFeeder.step: nothing uploaded")
        b = self.stepped % self.depth
        self.ready[b].synchronize()                       # host-side: no barrier packet on the step's queue
        out = self.trainer.train_step_packed(self.ring[b])
        self.done[b].record()
        self.stepped += 1
        return out


def synthetic_pairs(batch: int, eeg_channels: int = 64, samples: int = 1024, vol=(32, 32, 32),
                    seed: int = 1234, device="cuda", latent: int = 16):
    """pairs share a 16-d latent so retrieval is learnable:
    EEG = A_e z broadcast over time + N(0,1), fMRI = A_f z reshaped + N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    gm = torch.Generator().manual_seed(99)
    A_e = torch.randn(eeg_channels, latent, generator=gm)
    A_f = torch.randn(vol[0] * vol[1] * vol[2], latent, generator=gm) / latent ** 0.5
    z = torch.randn(batch, latent, generator=g)
    eeg = (z @ A_e.t()).unsqueeze(-1) * 0.5 + torch.randn(batch, eeg_channels, samples, generator=g)
    fmri = (z @ A_f.t()).view(batch, 1, *vol) + torch.randn(batch, 1, *vol, generator=g)
    return eeg.to(device), fmri.to(device)

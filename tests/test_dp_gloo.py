"""CPU, world_size 2, gloo: the three exchange steps of the data-parallel
contrastive step (multimodal_eeg_fmri_amd/dp.py) reproduce the single-process
global-batch computation.  The per-rank loss math is done here with the CPU
oracle standing in for the HIP kernel (which needs a GPU); what is under test is
the distributed algebra the GPU path uses verbatim: row offsets, "every rank
evaluates all rows of the gathered batch, so no reduce-scatter of column gradients
is needed" (mm_clip_loss_own_rows), the 1/world gradient scaling, and the split of
the flat gradient bucket into separately all-reduced parts."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_functional as RF


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, N, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_eeg_fmri_amd import dp
    try:
        g = torch.Generator().manual_seed(5)
        h_all = torch.randn(world * B, 2 * N, generator=g)          # pre-normalisation features, all ranks
        w = torch.randn(2 * N, generator=g)                          # a shared "parameter"
        scale = torch.tensor(3.0)
        h = (h_all[rank * B:(rank + 1) * B] * w).requires_grad_(True)
        z = torch.cat([RF.l2_normalize(h[:, :N]), RF.l2_normalize(h[:, N:])], dim=1)
        # --- the DP step exactly as ClipLossFn / BridgeTrainer do it: all-gather, then every rank evaluates
        # ALL rows of the gathered batch and keeps the gradient rows of its own pairs
        z_all = dp.gather_embeddings(z.detach(), None if world == 1 else dist.group.WORLD)
        assert z_all.shape == (world * B, 2 * N)
        za = z_all.clone().requires_grad_(True)
        total = 0.0
        for r in range(world):                                       # sum over ranks of their (local-mean) losses
            zr = za[r * B:(r + 1) * B]
            lr_ = RF.clip_loss(zr[:, :N], za[:, N:], za[:, :N], zr[:, N:], scale, row0=r * B)[0]
            total = total + lr_
            if r == rank:
                loss_r = lr_
        total.backward()
        dz_local = za.grad[rank * B:(rank + 1) * B]                  # what mm_clip_loss_own_rows writes
        z.backward(dz_local)                                         # into this rank's "encoder"
        gw = (h.grad * h_all[rank * B:(rank + 1) * B]).sum(0)        # d/dw on this rank
        flat = gw.clone()
        # the flat bucket goes out in one asynchronous piece per finished layer group (bridge_trainer._reduce_group: the
        # fMRI tail first, conv block 1's head of the bucket last), waited for only by the optimizer; the pieces are
        # disjoint views that tile the bucket, so the result is the one all-reduce
        n = flat.numel()
        cuts = [0, n // 8, n // 3, 2 * n // 3, n]
        works = [dp.allreduce_sum_(flat[lo:hi], dist.group.WORLD, async_op=True)
                 for lo, hi in reversed(list(zip(cuts[:-1], cuts[1:])))]
        assert all(wk is not None for wk in works)
        for wk in works:
            dp.wait(wk)
        flat /= world                                                # grad_scale = 1/world in the AdamW kernel
        losses = [torch.zeros(()) for _ in range(world)]
        dist.all_gather(losses, loss_r.detach())
        if rank == 0:
            out_q.put((flat, torch.stack(losses).mean()))
    finally:
        dist.destroy_process_group()


def test_dp_contrastive_step_equals_global_batch():
    world, B, N = 2, 6, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    got_grad, got_loss = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the global batch
    g = torch.Generator().manual_seed(5)
    h_all = torch.randn(world * B, 2 * N, generator=g)
    w = torch.randn(2 * N, generator=g).requires_grad_(True)
    h = h_all * w
    ze, zf = RF.l2_normalize(h[:, :N]), RF.l2_normalize(h[:, N:])
    loss = RF.clip_loss(ze, zf, ze, zf, torch.tensor(3.0))[0]
    loss.backward()
    assert abs(got_loss.item() - loss.item()) < 1e-6
    torch.testing.assert_close(got_grad, w.grad, rtol=1e-5, atol=1e-6)


def test_trainer_bucket_puts_the_fmri_slice_last():
    """bridge_trainer: the gradient bucket goes out in two parts - [fmri_lo, n) as soon as the fMRI branch's backward is
    done, [0, fmri_lo) after the EEG chain.  The fMRI encoder's parameters must be exactly the bucket's tail, and the two
    ranges must tile the bucket (host logic; no kernels run)."""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer
    tr = BridgeTrainer(eeg_channels=8, device="cpu", mode="manual")
    b = tr.bucket
    n_f = sum(p.numel() for p in tr.fmri_encoder.parameters())
    assert tr.fmri_lo == b.n - n_f and 0 < tr.fmri_lo < b.n
    base = b.g.data_ptr()
    for p in tr.fmri_encoder.parameters():
        off = (p._mm_grad.data_ptr() - base) // 4
        assert tr.fmri_lo <= off and off + p.numel() <= b.n
    for p in list(tr.eeg_encoder.parameters()) + [tr.head.logit_scale]:
        off = (p._mm_grad.data_ptr() - base) // 4
        assert off + p.numel() <= tr.fmri_lo
    assert b.g[:tr.fmri_lo].numel() + b.g[tr.fmri_lo:].numel() == b.n


def test_trainer_bucket_is_laid_out_by_finished_layer_group():
    """one contiguous range of the flat bucket per layer group, in the reverse of the order in which the backward finishes
    them: conv block 1 | conv blocks 2-3 | transformer stack + encoder head + projection heads + logit scale | fMRI
    encoder; the ranges tile the bucket, every parameter's gradient sink lies inside its group's range, and each group
    names the point of the backward at which it is reduced (host logic; no kernels run)."""
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer
    tr = BridgeTrainer(eeg_channels=8, device="cpu", mode="manual")
    b = tr.bucket
    assert [g[1] for g in tr.groups] == ["main", "handed1", "handed0", "fmri"]
    assert tr.groups[0][2] == 0 and tr.groups[-1][3] == b.n
    assert all(a[3] == c[2] for a, c in zip(tr.groups, tr.groups[1:]))            # contiguous, no gap, no overlap
    assert all(hi > lo for _, _, lo, hi in tr.groups)
    base = b.g.data_ptr()
    cl = tr.eeg_encoder.conv_layers
    members = {
        "main": list(cl[0].parameters()) + list(cl[1].parameters()),
        "handed1": [p for i in (4, 5, 9, 10) for p in cl[i].parameters()],
        "handed0": list(tr.eeg_encoder.transformer_layers.parameters()) + list(tr.eeg_encoder.output_proj.parameters())
        + list(tr.head.bridge.eeg_proj.parameters()) + list(tr.head.bridge.fmri_proj.parameters()) + [tr.head.logit_scale],
        "fmri": list(tr.fmri_encoder.parameters()),
    }
    total = 0
    for name, ready, lo, hi in tr.groups:
        for p in members[ready]:
            off = (p._mm_grad.data_ptr() - base) // 4
            assert lo <= off and off + p.numel() <= hi, (name, off, lo, hi)
        assert sum(p.numel() for p in members[ready]) == hi - lo, name
        total += hi - lo
    assert total == b.n


def test_trainer_accepts_the_config5_eeg_branch():
    """BridgeTrainer(eeg_encoder=...): the STFT front-end + power encoder (BASELINE config #5) and a bare EnhancedPowerEncoder
    as the EEG branch get a two-group bucket (EEG side | fMRI encoder); an encoder without a tape is refused (host logic)."""
    import torch.nn as nn
    from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer
    from multimodal_eeg_fmri_amd.crossmodal_v4_enhancements import MultiScaleSTFTPowerEncoder
    from multimodal_eeg_fmri_amd.enhanced_models_v4 import EnhancedPowerEncoder
    for enc, kind in ((MultiScaleSTFTPowerEncoder(4, (16, 32), 8, 128, 1, 4, 0.1), "stft"), (EnhancedPowerEncoder(8, 128, 1, 4, 0.1), "power")):
        tr = BridgeTrainer(device="cpu", mode="manual", eeg_encoder=enc)
        assert tr._eeg_kind == kind and [g[1] for g in tr.groups] == ["main", "fmri"]
        assert tr.groups[0][2] == 0 and tr.groups[0][3] == tr.groups[1][2] == tr.fmri_lo and tr.groups[1][3] == tr.bucket.n
        assert all(getattr(p, "_mm_grad", None) is not None for p in enc.parameters())
    with pytest.raises(TypeError):
        BridgeTrainer(device="cpu", eeg_encoder=nn.Linear(4, 128))


def _agree_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_eeg_fmri_amd import dp
    try:
        out = []
        g = dist.group.WORLD
        out.append(dp.agree_on_capture(True, 5, g))                       # every rank captured
        out.append(dp.agree_on_capture(False, 0, g))                      # every rank refused before any collective
        for ok, enq in (((rank == 0), 5 if rank == 0 else 0),             # one rank failed, cleanly - but its peer captured
                        (False, 1 if rank == 1 else 0)):                  # all failed, one of them after a collective
            out.append(dp.agree_on_capture(ok, enq, g))
        dp.check_communicator(g, "cpu")                                   # a working group passes
        real = dist.all_reduce

        def short_sum(t, *a, **k):                                        # a communicator that lost a rank's contribution
            real(t, *a, **k)
            t.sub_(1.0)
        dist.all_reduce = short_sum
        try:
            dp.check_communicator(g, "cpu")
            out.append("no error")
        except RuntimeError as e:
            out.append("raised" if "aborting the job" in str(e) else str(e))
        finally:
            dist.all_reduce = real
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_the_capture_form():
    """the fallback from the one-graph step to the segmented form must be a GROUP decision.  All captured ->
    "captured"; all refused before a collective was recorded -> "segments" on every rank; anything mixed (a rank failed
    while another captured, or a failure after a collective had been recorded) -> "segments-after-abort" on EVERY rank
    (recorded collectives never ran: all drop their graphs together), after which the communicator has to pass a
    known-answer all-reduce - a wrong sum raises on every rank, so the job exits non-zero."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r] == ["captured", "segments", "segments-after-abort", "segments-after-abort", "raised"], got
    from multimodal_eeg_fmri_amd import dp
    assert dp.agree_on_capture(True, 3, None) == "captured" and dp.agree_on_capture(False, 0, None) == "segments"
    assert dp.agree_on_capture(False, 2, None) == "segments-after-abort"

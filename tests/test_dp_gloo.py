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
        # the flat bucket goes out in parts (bridge_trainer: the branch that finishes first is reduced
        # while the other still runs); the parts are disjoint views, so the result is the one all-reduce
        cut = flat.numel() // 3
        works = [dp.allreduce_sum_(flat[:cut], dist.group.WORLD, async_op=True),
                 dp.allreduce_sum_(flat[cut:], dist.group.WORLD, async_op=True)]
        for wk in works:
            wk.wait()
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

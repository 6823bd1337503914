"""CPU, world_size 2, gloo: the three exchange steps of the data-parallel
contrastive step (multimodal_eeg_fmri_amd/dp.py) reproduce the single-process
global-batch computation.  The per-rank loss math is done here with the CPU
oracle standing in for the HIP kernel (which needs a GPU); what is under test is
the distributed algebra the GPU path uses verbatim: row offsets, the
reduce-scatter of column gradients and the 1/world gradient scaling."""
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
        # --- the DP step exactly as ClipLossFn does it
        z_all = dp.gather_embeddings(z.detach(), None if world == 1 else dist.group.WORLD)
        assert z_all.shape == (world * B, 2 * N)
        za = z_all.clone().requires_grad_(True)
        zl = za[rank * B:(rank + 1) * B]
        loss_r = RF.clip_loss(zl[:, :N], za[:, N:], za[:, :N], zl[:, N:], scale, row0=rank * B)[0]
        loss_r.backward()                                            # d loss_r / d z_all (row + column roles)
        dz_local = dp.scatter_column_grads(za.grad, dist.group.WORLD)
        z.backward(dz_local)                                         # into this rank's "encoder"
        gw = (h.grad * h_all[rank * B:(rank + 1) * B]).sum(0)        # d/dw on this rank
        flat = gw.clone()
        dp.allreduce_sum_(flat, dist.group.WORLD)
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

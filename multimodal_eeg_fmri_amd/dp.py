"""Data-parallel exchange steps of the contrastive bridge (backend-agnostic:
RCCL on the GPUs, gloo in the CPU tests).  SURVEY.md section 8e lists three exchanges per step:

  1. all-gather of the packed L2-normalised embeddings  (global negatives)
  2. reduce-scatter (sum) of the gradients w.r.t. the gathered embeddings
  3. all-reduce of the flat fp32 gradient bucket (mean is applied as
     ``grad_scale = 1/world`` inside the fused AdamW kernel)

The trainer's tape (bridge_trainer.py) needs only 1 and 3: every rank evaluates all rows of the gathered
batch (``mm_clip_loss_own_rows``) and so already holds the sum step 2 would deliver for its own rows.
Step 2 remains for the public autograd surface (``ops.clip_loss`` with a process group).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size(group) -> int:
    return dist.get_world_size(group) if group is not None else 1


def rank(group) -> int:
    return dist.get_rank(group) if group is not None else 0


def _host_staged(t: torch.Tensor, group) -> bool:
    """gloo carrying device tensors (the one-GPU multi-rank rehearsal, tests only):
    stage through host memory.  RCCL (backend "nccl") never takes this branch."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_into(out: torch.Tensor, z_local: torch.Tensor, group) -> torch.Tensor:
    if _host_staged(out, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, z_local.detach().cpu().contiguous(), group=group)
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, z_local.contiguous(), group=group)
    return out


def reduce_scatter_into(out: torch.Tensor, dz_all: torch.Tensor, group) -> torch.Tensor:
    if _host_staged(out, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(h, dz_all.detach().cpu().contiguous(), op=dist.ReduceOp.SUM, group=group)
        out.copy_(h)
    else:
        dist.reduce_scatter_tensor(out, dz_all.contiguous(), op=dist.ReduceOp.SUM, group=group)
    return out


def gather_embeddings(z_local: torch.Tensor, group) -> torch.Tensor:
    """(B, 2N) per rank -> (world*B, 2N), rank r's rows at [r*B, (r+1)*B)."""
    w = world_size(group)
    if w == 1:
        return z_local
    out = torch.empty((w * z_local.shape[0], z_local.shape[1]), dtype=z_local.dtype, device=z_local.device)
    return all_gather_into(out, z_local, group)


def scatter_column_grads(dz_all: torch.Tensor, group) -> torch.Tensor:
    """sum over ranks of d(loss_r)/d(z_all), returning this rank's row block."""
    w = world_size(group)
    if w == 1:
        return dz_all
    B = dz_all.shape[0] // w
    out = torch.empty((B, dz_all.shape[1]), dtype=dz_all.dtype, device=dz_all.device)
    return reduce_scatter_into(out, dz_all, group)


def allreduce_sum_(flat: torch.Tensor, group) -> torch.Tensor:
    if world_size(group) > 1:
        if _host_staged(flat, group):
            h = flat.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(h)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat

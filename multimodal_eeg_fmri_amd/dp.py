"""Data-parallel exchange steps of the contrastive bridge (RCCL on the GPUs, gloo on CPU tensors in the
CPU tests).  SURVEY.md section 8e lists three exchanges per step:

  1. all-gather of the packed L2-normalised embeddings  (global negatives)
  2. reduce-scatter (sum) of the gradients w.r.t. the gathered embeddings
  3. all-reduce of the flat fp32 gradient bucket (mean is applied as
     ``grad_scale = 1/world`` inside the fused AdamW kernel)

Only 1 and 3 are issued: every rank evaluates all rows of the gathered batch
(``mm_clip_loss_own_rows``) and so already holds the sum step 2 would deliver for its own rows
(tests/test_dp_gloo.py checks that algebra at world size 2).

The one-GPU multi-rank rehearsal (gloo carrying device tensors through host memory) lives in
``tools/gloo_staging.py``; it replaces the three functions below from outside and is not product code.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


# the collectives below can be recorded into a hipGraph (RCCL: yes).  tools/gloo_staging.py (one-GPU rehearsals: gloo
# carrying device tensors through host memory) sets this to False, and the trainer then keeps its three-segment form.
CAPTURABLE = True


def world_size(group) -> int:
    return dist.get_world_size(group) if group is not None else 1


def rank(group) -> int:
    return dist.get_rank(group) if group is not None else 0


def all_gather_into(out: torch.Tensor, z_local: torch.Tensor, group) -> torch.Tensor:
    dist.all_gather_into_tensor(out, z_local.contiguous(), group=group)
    return out


def gather_embeddings(z_local: torch.Tensor, group) -> torch.Tensor:
    """(B, 2N) per rank -> (world*B, 2N), rank r's rows at [r*B, (r+1)*B)."""
    w = world_size(group)
    if w == 1:
        return z_local
    out = torch.empty((w * z_local.shape[0], z_local.shape[1]), dtype=z_local.dtype, device=z_local.device)
    return all_gather_into(out, z_local, group)


def allreduce_sum_(flat: torch.Tensor, group, async_op: bool = False):
    """in-place sum over ranks; ``async_op`` returns the work handle (None at world 1)"""
    if world_size(group) > 1:
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op) if async_op \
            else (dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group), None)[1]
    return None

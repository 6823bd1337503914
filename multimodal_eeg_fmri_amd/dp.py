"""Data-parallel exchange steps of the contrastive bridge (RCCL on the GPUs, gloo on CPU tensors in the
CPU tests).  SURVEY.md section 8e lists three exchanges per step:

  1. all-gather of the packed L2-normalised embeddings  (global negatives)
  2. reduce-scatter (sum) of the gradients w.r.t. the gathered embeddings
  3. all-reduce of the flat fp32 gradient bucket (mean is applied as
     ``grad_scale = 1/world`` inside the fused AdamW kernel)

Only 1 and 3 are issued: every rank evaluates all rows of the gathered batch
(``mm_clip_loss_own_rows``) and so already holds the sum step 2 would deliver for its own rows
(tests/test_dp_gloo.py checks that algebra at world size 2).  Step 3 goes out in one piece per finished
layer group of the backward (``bridge_trainer.BridgeTrainer.groups``), each as an asynchronous collective
whose completion only the optimizer waits for.

The one-GPU multi-rank rehearsal (gloo carrying device tensors through host memory) lives in
``tools/gloo_staging.py``; it replaces the exchange functions below from outside and is not product code.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# the collectives below can be recorded into a hipGraph (RCCL: yes).  tools/gloo_staging.py (one-GPU rehearsals: gloo
# carrying device tensors through host memory) sets this to False, and the trainer then keeps its three-segment form.
CAPTURABLE = True
# MM_DP_FORCE=1 (rehearsal on one GPU): a group of ONE rank still issues every collective through the backend, so that the
# all-reduce nodes of the captured step exist and are replayed (a world of one otherwise skips them: nothing to sum)
FORCE_COLLECTIVES = bool(os.environ.get("MM_DP_FORCE"))
# collectives handed to the backend so far by this process (the capture agreement asks whether an aborted capture had
# already enqueued one: the communicator's state is then undefined)
issued = 0


def world_size(group) -> int:
    return dist.get_world_size(group) if group is not None else 1


def rank(group) -> int:
    return dist.get_rank(group) if group is not None else 0


def active(group) -> bool:
    """the step issues collectives on this group"""
    return group is not None and (world_size(group) > 1 or FORCE_COLLECTIVES)


def all_gather_into(out: torch.Tensor, z_local: torch.Tensor, group) -> torch.Tensor:
    global issued
    dist.all_gather_into_tensor(out, z_local.contiguous(), group=group)
    issued += 1
    return out


def gather_embeddings(z_local: torch.Tensor, group) -> torch.Tensor:
    """(B, 2N) per rank -> (world*B, 2N), rank r's rows at [r*B, (r+1)*B)."""
    w = world_size(group)
    if w == 1 and not active(group):
        return z_local
    out = torch.empty((w * z_local.shape[0], z_local.shape[1]), dtype=z_local.dtype, device=z_local.device)
    return all_gather_into(out, z_local, group)


def allreduce_sum_(flat: torch.Tensor, group, async_op: bool = False):
    """in-place sum over ranks; ``async_op`` returns the work handle (None when nothing was issued): the caller's
    stream does not wait for the collective until ``wait(handle)``"""
    global issued
    if not active(group):
        return None
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    issued += 1
    return work if async_op else None


def wait(work) -> None:
    """the current stream waits for an asynchronous collective (no host block with RCCL; capturable)"""
    if work is not None:
        work.wait()


# ------------------------------------------------------------------ host-side agreement
_CTRL = {}


def _control_group(group):
    """a gloo group over the ranks of ``group`` for host-side agreement: it does not touch the RCCL communicator,
    whose state may be the very thing in question.  Collective: every rank of ``group`` must call it."""
    key = id(group)
    if key not in _CTRL:
        if dist.get_backend(group) == "gloo":
            _CTRL[key] = group
        else:
            ranks = dist.get_process_group_ranks(group)
            _CTRL[key] = dist.new_group(ranks=ranks, backend="gloo")
    return _CTRL[key]


def agree_on_capture(ok: bool, issued_in_attempt: int, group) -> str:
    """every rank reports whether its capture of the step succeeded -> ONE verdict for the whole group, so that no two
    ranks ever replay different collective sequences:

    * ``"captured"``  every rank holds the one-graph step;
    * ``"segments"``  every rank was refused before any collective had been handed to the backend: all fall back to
      the segmented form together;
    * ``"segments-after-abort"``  some rank failed while another captured, or a capture was aborted after collectives
      had been recorded.  Recorded collectives never ran (a capture executes nothing), so all ranks drop their graphs
      and fall back together - but the communicator has seen an aborted capture, and the caller must prove it still
      works (``check_communicator``) before the first step."""
    if group is None or world_size(group) == 1:
        return "captured" if ok else ("segments" if issued_in_attempt == 0 else "segments-after-abort")
    t = torch.tensor([0 if ok else 1, 0 if ok else int(issued_in_attempt)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_control_group(group))
    failed, enq = int(t[0]), int(t[1])
    if failed == 0:
        return "captured"
    if failed == world_size(group) and enq == 0:
        return "segments"
    return "segments-after-abort"


def check_communicator(group, device) -> None:
    """one eager all-reduce over the data group whose answer is known (ones -> world size).  After an aborted capture
    this is what tells a working communicator from a broken one: a wrong sum raises here on every rank; a collective
    that never completes is ended by the process group's own timeout (the job exits non-zero, it does not hang)."""
    w = world_size(group)
    t = torch.ones(64, dtype=torch.float32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    got = t.cpu()
    if not bool((got == float(w)).all()):
        raise RuntimeError(f"the data-parallel communicator returns {got[:4].tolist()} for a sum of ones over {w} rank(s) after an "
                           "aborted hipGraph capture; aborting the job (set MM_DP_CAPTURE=0 for the segmented form)")

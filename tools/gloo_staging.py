"""Rehearsal shim (NOT product code): lets several ranks share ONE GPU by carrying device tensors over
the gloo backend through host memory.  RCCL (backend "nccl") refuses two ranks on one device, so the
one-GPU rehearsals of the N > 1 step (tools/dp_rehearsal.py, the MM_DIST_BACKEND=gloo leg of bench.py used by
tests/test_trainer_gpu.py) call ``install()`` to replace the exchange functions of
``multimodal_eeg_fmri_amd.dp`` from outside; the product module itself only issues the plain collectives."""
import torch
import torch.distributed as dist


def install():
    from multimodal_eeg_fmri_amd import dp

    def all_gather_into(out, z_local, group):
        if not out.is_cuda:
            dist.all_gather_into_tensor(out, z_local.contiguous(), group=group)
            return out
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, z_local.detach().cpu().contiguous(), group=group)
        out.copy_(h)
        return out

    def allreduce_sum_(flat, group, async_op=False):
        if dp.world_size(group) > 1:
            if flat.is_cuda:
                h = flat.detach().cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                flat.copy_(h)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return None

    dp.all_gather_into = all_gather_into
    dp.allreduce_sum_ = allreduce_sum_
    dp.CAPTURABLE = False                 # host round trips cannot be recorded into a hipGraph

"""Data-parallel plumbing: one process per GPU, images sharded across ranks, final boxes gathered.

Images are independent end to end (eval-mode BN, per-image decode -- SURVEY.md section 8e), so the
only exchange is the gather of the fixed-size detection records after decode.  The reference has no
counterpart (its torch.distributed imports at train.py:11,17 are unused).  The collective goes
through ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests); payloads are a few hundred KB per rank, so it is latency- not bandwidth-bound.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of ``n_items`` for ``rank``; the first ``n_items % world`` ranks
    get one extra item.  Concatenating shards by rank reproduces the single-process order."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


REC = 16   # floats per detection record: x1,y1,x2,y2,score,cls, lm0..lm9


def pack_records(dets, lms):
    """[B,K,6] + [B,K,10] -> [B,K,16] (numpy or torch; same array namespace as the inputs)."""
    if hasattr(dets, "new_empty"):                       # torch
        import torch
        return torch.cat([dets, lms], dim=2)
    return np.concatenate([dets, lms], axis=2)


def gather_records(local, group=None):
    """All-gather equal-shape per-rank record tensors [b,K,16] into [world*b,K,16], rank-major,
    i.e. exactly the batch order of the unsharded run.  ``local`` is a torch tensor on the device
    the process group's backend expects (cuda for nccl, cpu for gloo)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out

"""Data-parallel plumbing: one process per GPU, images sharded across ranks, final boxes gathered.

Images are independent end to end (eval-mode BN, per-image decode -- SURVEY.md section 8e), so the
only exchange is the gather of the fixed-size detection records after decode.  The reference has no
counterpart (its torch.distributed imports at train.py:11,17 are unused).  Two equivalent gathers:

* ``Comm`` / ``Comm.gather_topk``: the C ABI's ``cf_gather_topk`` -- decode + ``ncclAllGather`` (RCCL over xGMI)
  enqueued on the context's decode stream, no Python or torch on the data path; what a C / cgo / JNI host uses too.
  The 128-byte RCCL id is shipped by the caller (``unique_id`` / ``broadcast_unique_id`` over torch.distributed's
  store, or any other channel).
* ``gather_records``: ``torch.distributed.all_gather_into_tensor`` (backend "nccl" = RCCL on the GPU box, "gloo"
  in the CPU tests) on records packed by ``pack_records``.

Payloads are a few hundred KB per rank, so the gather is latency- not bandwidth-bound.
"""
import ctypes as C

import numpy as np

from . import _lib


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


COMM_ID_BYTES = 128


def unique_id():
    """A fresh RCCL unique id (bytes, 128) -- call on ONE rank and ship it to the others."""
    buf = (C.c_char * COMM_ID_BYTES)()
    _lib.check(_lib.lib().cf_comm_unique_id(buf, COMM_ID_BYTES))
    return bytes(buf.raw)


def broadcast_unique_id(src=0, group=None):
    """Rank ``src`` creates the id, every rank of the (initialised) torch.distributed group receives it."""
    import torch.distributed as dist
    obj = [unique_id() if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(obj, src=src, group=group)
    return obj[0]


class Comm(object):
    """RCCL communicator bound to one Engine (one GPU): ``cf_comm_create`` / ``cf_gather_topk``."""

    def __init__(self, engine, rank, world, uid):
        if len(uid) != COMM_ID_BYTES:
            raise ValueError("RCCL unique id must be %d bytes" % COMM_ID_BYTES)
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        h = C.c_void_p()
        _lib.check(_lib.lib().cf_comm_create(engine._h, self.rank, self.world, C.c_char_p(uid), C.byref(h)), engine._h)
        self._h = h

    def gather_topk(self, K=100, use_reg=True):
        """Decode the engine's last forward and all-gather: float32 [world * B, K, 16] (host, blocking)."""
        B = self.engine.last_B
        out = np.empty((self.world * B, int(K), REC), np.float32)
        _lib.check(_lib.lib().cf_gather_topk(self.engine._h, self._h, int(K), 1 if use_reg else 0, _lib.ptr(out), 0), self.engine._h)
        return out

    def gather_topk_device(self, K, records_ptr, use_reg=True):
        """Same into a caller-owned DEVICE buffer [world * B, K, 16] (asynchronous, decode stream)."""
        _lib.check(_lib.lib().cf_gather_topk(self.engine._h, self._h, int(K), 1 if use_reg else 0, C.c_void_p(int(records_ptr)), 1),
                   self.engine._h)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cf_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

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


_calls = {}


def _group_tag(group):
    """Key namespace of a process group: the global ranks of its members (two disjoint subgroups bootstrapping at the same
    time must not share store keys) + a per-namespace call counter (a second call must not read the first one's values)."""
    import torch.distributed as dist
    ranks = list(range(dist.get_world_size())) if group is None else [int(r) for r in dist.get_process_group_ranks(group)]
    return "g" + "_".join(str(r) for r in ranks)


def _next_key(prefix):
    n = _calls.get(prefix, 0)
    _calls[prefix] = n + 1
    return "%s/%d" % (prefix, n)


def broadcast_unique_id(src=0, group=None, store=None):
    """Rank ``src`` (a rank OF ``group``, group-relative like torch.distributed.broadcast's ``group_src``) creates the id,
    every rank of the (initialised) torch.distributed group receives it -- through the key-value store (TCP), not through a
    collective: no NCCL communicator has to exist (or work) for the bootstrap of this one.  Collective in the sense that every
    rank of the group must call it the same number of times (the key carries the group's ranks and a call counter)."""
    import torch.distributed as dist
    store = store or dist.distributed_c10d._get_default_store()
    key = _next_key("cf_comm_uid/" + _group_tag(group))
    src_global = int(src) if group is None else int(dist.get_global_rank(group, int(src)))
    if dist.get_rank() == src_global:
        try:
            uid = unique_id()
        except Exception:
            store.set(key, b"failed")                # the other ranks must not wait for an id that will never come
            raise
        store.set(key, uid)
        return uid
    uid = bytes(store.get(key))                      # blocks until rank src has published it (store timeout applies)
    if len(uid) != COMM_ID_BYTES:
        raise RuntimeError("rank %d could not create the RCCL unique id" % src)
    return uid


class Comm(object):
    """The RCCL communicator of this rank (= this GPU): ``cf_comm_create`` / ``cf_gather_topk``.

    ONE per rank, shared by every Engine (context) the rank drives on that GPU: the communicator owns the rank's single
    gather stream and all-gathers are enqueued there in call order, so all ranks see the same collective order as long as
    they call ``gather_topk*`` in the same order.  ``engine`` names the device at creation and is the default context of
    the gather calls; pass ``engine=`` to gather another context's decode through the same communicator."""

    def __init__(self, engine, rank, world, uid, _handle=None):
        self.engine, self.rank, self.world = engine, int(rank), int(world)
        if _handle is not None:
            self._h = _handle
            return
        if len(uid) != COMM_ID_BYTES:
            raise ValueError("RCCL unique id must be %d bytes" % COMM_ID_BYTES)
        h = C.c_void_p()
        _lib.check(_lib.lib().cf_comm_create(engine._h, self.rank, self.world, C.c_char_p(uid), C.byref(h)), engine._h)
        self._h = h

    @classmethod
    def create_all(cls, engines):
        """One process that owns one Engine per GPU: all communicators in one grouped call (``cf_comm_create_all``);
        rank i = engines[i]."""
        n = len(engines)
        ctxs = (C.c_void_p * n)(*[e._h for e in engines])
        outs = (C.c_void_p * n)()
        _lib.check(_lib.lib().cf_comm_create_all(ctxs, n, outs), engines[0]._h)
        return [cls(e, i, n, None, _handle=C.c_void_p(outs[i])) for i, e in enumerate(engines)]

    @classmethod
    def loopback(cls, engine, world):
        """Dry run of a ``world``-rank gather on ONE GPU (``cf_comm_create_loopback``): no RCCL, this process plays the ranks in
        turn -- ``play(rank)`` before each ``gather_topk*``; the call of the step's last rank returns the gathered
        [world * B, K, 16] records, the calls before it return None (host form) / write nothing (device form).  Slot sizes,
        header protocol, step numbers and the rank-major unpack are those of the real gather."""
        h = C.c_void_p()
        _lib.check(_lib.lib().cf_comm_create_loopback(engine._h, int(world), C.byref(h)), engine._h)
        m = cls(engine, 0, world, None, _handle=h)
        m._loopback, m._played = True, set()
        return m

    def play(self, rank):
        """Loopback only: the rank whose gather comes next (every rank exactly once per step, any order)."""
        if _lib.lib().cf_comm_loopback_rank(self._h, int(rank)) != 0:
            raise ValueError("play(%r): not a loopback communicator, or rank outside [0, %d)" % (rank, self.world))
        self.rank = int(rank)

    def set_shard(self, B, K):
        """``cf_comm_set_shard``: declare the shard (B images x K records) every rank gathers.  Collective by contract (same
        values, same point of the call sequence on every rank); enqueues the one agreement collective of the gather path and
        returns without waiting -- ``wait(timeout)`` / ``query()`` give the verdict (RuntimeError on unequal shards).  The
        first ``gather_topk*`` calls it implicitly and then waits for the verdict itself."""
        r = _lib.lib().cf_comm_set_shard(self._h, int(B), int(K))
        if r != 0:
            raise RuntimeError("cf_comm_set_shard failed (%d): %s" % (r, self.last_error()))

    def gather_topk(self, K=100, use_reg=True, engine=None):
        """Decode the engine's last forward and all-gather: float32 [world * B, K, 16] (host, blocking)."""
        eng = engine or self.engine
        out = np.empty((self.world * eng.last_B, int(K), REC), np.float32)
        _lib.check(_lib.lib().cf_gather_topk(eng._h, self._h, int(K), 1 if use_reg else 0, _lib.ptr(out), 0), eng._h)
        if getattr(self, "_loopback", False):            # only the step's last rank holds the gathered records
            self._played.add(self.rank)
            if len(self._played) < self.world:
                return None
            self._played.clear()
        return out

    def gather_topk_device(self, K, records_ptr, use_reg=True, engine=None):
        """Same into a caller-owned DEVICE buffer [world * B, K, 16] (asynchronous: decode stream of the engine, then the
        communicator's gather stream; ``synchronize()`` or ``engine.synchronize()`` before reading)."""
        eng = engine or self.engine
        _lib.check(_lib.lib().cf_gather_topk(eng._h, self._h, int(K), 1 if use_reg else 0, C.c_void_p(int(records_ptr)), 1), eng._h)

    def query(self):
        """True while an enqueued gather is still running (never blocks)."""
        r = _lib.lib().cf_comm_query(self._h)
        if r < 0:
            raise RuntimeError("cf_comm_query failed (%d): %s" % (r, self.last_error()))
        return r == 1

    def last_error(self):
        return (_lib.lib().cf_comm_last_error(self._h) or b"").decode()

    def debug(self, what, value):
        """Test hooks of ``cf_comm_debug``: (0, ms) parks the gather stream behind a spin kernel, (1, d) skews the B in the header of the
        next gather's slot."""
        if _lib.lib().cf_comm_debug(self._h, int(what), int(value)) != 0:
            raise RuntimeError("cf_comm_debug(%d, %d) failed" % (what, value))

    def wait(self, timeout_s, poll_s=0.002):
        """Poll ``query`` until the gather stream is idle; False if it is still busy after ``timeout_s`` seconds
        (a collective that does not complete: abort() instead of hanging in a synchronize)."""
        import time
        t0 = time.perf_counter()
        while self.query():
            if time.perf_counter() - t0 > timeout_s:
                return False
            time.sleep(poll_s)
        return True

    def synchronize(self):
        r = _lib.lib().cf_comm_synchronize(self._h)
        if r != 0:
            raise RuntimeError("cf_comm_synchronize failed (%d): %s" % (r, self.last_error()))

    def stream(self):
        return int(_lib.lib().cf_comm_stream(self._h) or 0)

    def abort(self):
        """ncclCommAbort + release: for a communicator whose collective hangs.  The object is dead afterwards."""
        if getattr(self, "_h", None):
            _lib.lib().cf_comm_abort(self._h)
            self._h = None

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cf_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def agree(ok, rank, world, key, store=None):
    """Every rank publishes ``ok`` under ``key`` in the torch.distributed key-value store (TCP, NOT a NCCL collective: it
    works while a RCCL collective is hung) and reads everybody's: True only if all ranks said ok.  Every rank must call it the
    same number of times per key (the store keys carry a call counter, so a repeat never reads an earlier call's verdicts)."""
    if world == 1:
        return bool(ok)
    import torch.distributed as dist
    store = store or dist.distributed_c10d._get_default_store()
    k = _next_key("cf_agree/%s/w%d" % (key, world))
    store.set("%s/%d" % (k, rank), b"1" if ok else b"0")
    return all(store.get("%s/%d" % (k, r)) == b"1" for r in range(world))

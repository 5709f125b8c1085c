"""The N > 1 path before the driver runs it on 8 GPUs: the exact `step()` of bench.py on one GPU (C-ABI RCCL
gather at world 1 and the torch.distributed variant), and a world-size-2 run (two processes, both on GPU 0, gloo
rendezvous, each with its own Engine on its shard) whose gathered result must equal the unsharded result BITWISE
(SURVEY.md section 8e).  RCCL itself refuses two ranks on one device, so the 2-rank run gathers through gloo; the
RCCL collective is exercised at world 1 (communicator creation, decode-stream all-gather, record layout)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import centerface_amd as cfa

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(eng, K):
    d, l, _ = eng.decode_topk(K)
    return cfa.distributed.pack_records(d, l)


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("gather", ["cf", "torch"])
def test_bench_multi_gpu_step_on_one_gpu(gather, depth):
    """bench.py's step for N > 1 (identity gather at world 1), on one context and on the ring of two contexts it uses
    by default: the records of every step must equal a plain decode of the same batch."""
    import torch
    import bench
    B, S, K = 8, 160, 50
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    imgs = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    d_in = torch.from_numpy(imgs).to(dev)
    outs = [{"dets": torch.empty((B, K, 6), dtype=torch.float32, device=dev), "lms": torch.empty((B, K, 10), dtype=torch.float32, device=dev),
             "inds": torch.empty((B, K), dtype=torch.int64, device=dev), "all": torch.zeros((B, K, 16), dtype=torch.float32, device=dev)}
            for _ in range(depth)]
    engs = [cfa.Engine(S, S, max_batch=B, dtype="bf16") for _ in range(depth)]
    # ONE communicator per rank, shared by the rank's contexts (its single gather stream orders the collectives)
    comms = [cfa.distributed.Comm(engs[0], 0, 1, cfa.distributed.unique_id())] if gather == "cf" else None
    step = bench.make_step(cfa, engs, d_in.data_ptr(), B, K, outs, gather, comms)
    got = [step() for _ in range(4 * depth)]             # per context: eager, capture, replay, replay; gathers overlap the next forward
    for e in engs:
        e.synchronize()
    if comms is not None:
        assert comms[0].wait(30.0) and not comms[0].query()
    torch.cuda.synchronize()
    got = [g.cpu().numpy() for g in got[-depth:]]        # the last result of every context
    engs[0].forward_enqueue(imgs)
    want = _records(engs[0], K)
    for g in got:
        assert np.array_equal(g, want)
    if comms is not None:                                # host-destination variant of the same entry point, through every context
        for e in engs:
            e.forward_enqueue(imgs)
            assert np.array_equal(comms[0].gather_topk(K, engine=e), want)
        for cm in comms:
            cm.close()
    for e in engs:
        e.close()


def test_bench_step_configs4_shard_two_contexts_one_communicator():
    """The BASELINE configs[4] per-GPU shard (1280x1280, top-1000, B = 4: the `topk_select_kernel<1024, *>` + gather-record
    path) through bench.py's N > 1 step: two contexts sharing the rank's one communicator / gather stream."""
    import torch
    import bench
    B, S, K, depth = 4, 1280, 1000, 2
    dev = torch.device("cuda", 0)
    imgs = np.random.default_rng(12).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    d_in = torch.from_numpy(imgs).to(dev)
    outs = [{"dets": torch.empty((B, K, 6), dtype=torch.float32, device=dev), "lms": torch.empty((B, K, 10), dtype=torch.float32, device=dev),
             "inds": torch.empty((B, K), dtype=torch.int64, device=dev), "all": torch.zeros((B, K, 16), dtype=torch.float32, device=dev)}
            for _ in range(depth)]
    engs = [cfa.Engine(S, S, max_batch=B, dtype="bf16") for _ in range(depth)]
    comm = cfa.distributed.Comm(engs[0], 0, 1, cfa.distributed.unique_id())
    step = bench.make_step(cfa, engs, d_in.data_ptr(), B, K, outs, "cf", [comm])
    got = [step() for _ in range(3 * depth)]
    assert comm.wait(60.0)
    for e in engs:
        e.synchronize()
    got = [g.cpu().numpy() for g in got[-depth:]]
    engs[0].forward_enqueue(imgs)
    want = _records(engs[0], K)
    for g in got:
        assert np.array_equal(g, want)
    comm.close()
    for e in engs:
        e.close()


@pytest.mark.isolated
def test_two_contexts_in_flight_are_deterministic():
    """Two contexts running concurrently on one GPU (the benchmarked schedule) return, step after step, exactly what one context
    returns alone.  Regression test for the LDS-DMA publication race (a wave passing the barrier before another wave's
    `global_load_lds` had landed: rare stale expand weights, only when a second context competed for the chip)."""
    import torch
    B, S, K = 8, 640, 100
    dev = torch.device("cuda", 0)
    imgs = np.random.default_rng(21).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    d_in = torch.from_numpy(imgs).to(dev)
    fmt = cfa._lib.CF_IN_U8_HWC_BGR
    engs = [cfa.Engine(S, S, max_batch=B, dtype="bf16") for _ in range(2)]
    engs[0].forward_enqueue(imgs)
    want = engs[0].decode_topk(K)
    outs = [[torch.empty((B, K, 6), dtype=torch.float32, device=dev), torch.empty((B, K, 10), dtype=torch.float32, device=dev),
             torch.empty((B, K), dtype=torch.int64, device=dev)] for _ in engs]
    for it in range(24):
        e, o = engs[it % 2], outs[it % 2]
        if it >= 2:                                          # the result this slot produced two steps ago
            e.synchronize()
            assert np.array_equal(o[0].cpu().numpy(), want[0]) and np.array_equal(o[2].cpu().numpy(), want[2]), it
        e.forward_enqueue(d_in.data_ptr(), on_device=True, B=B, in_format=fmt)
        e.decode_topk_device(K, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    for e in engs:
        e.close()


@pytest.mark.isolated
def test_comm_create_all_grouped_and_abort():
    """cf_comm_create_all (the single-process host: n ncclCommInitRank calls inside one ncclGroupStart/End) at n = 1, its
    gather, and cf_comm_abort on a live communicator; two contexts on one device are refused (one rank per GPU)."""
    S, B, K = 96, 3, 20
    imgs = np.random.default_rng(13).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    eng = cfa.Engine(S, S, max_batch=B, dtype="fp32")
    comms = cfa.distributed.Comm.create_all([eng])
    assert len(comms) == 1 and comms[0].world == 1 and comms[0].stream() != 0
    eng.forward_enqueue(imgs)
    rec = comms[0].gather_topk(K)
    eng.forward_enqueue(imgs)
    assert np.array_equal(rec, _records(eng, K))
    comms[0].abort()
    comms[0].abort()                                     # idempotent
    eng2 = cfa.Engine(S, S, max_batch=B, dtype="fp32")
    with pytest.raises((RuntimeError, ValueError)):
        cfa.distributed.Comm.create_all([eng, eng2])
    eng.close(); eng2.close()


def test_bench_gather_fallback_records_the_path():
    """bench.py --exercise-gather-path with a zero gather deadline: the warm-up gather "times out", the communicator is
    aborted, the run continues on the torch.distributed path with one context and the JSON line says so."""
    import json
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "2",
                          "--batch", "4", "--size", "160", "--topk", "20", "--no-cpu-baseline", "--no-extras", "--profile-reps", "1",
                          "--exercise-gather-path", "--gather-timeout", "-1"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["config"]["gather_fallback"] and "torch.distributed" in d["config"]["gather"] and d["config"]["contexts_per_gpu"] == 1
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "2",
                          "--batch", "4", "--size", "160", "--topk", "20", "--no-cpu-baseline", "--no-extras", "--profile-reps", "1",
                          "--exercise-gather-path"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["config"]["gather_fallback"] is None and "cf_gather_topk" in d["config"]["gather"] and d["config"]["contexts_per_gpu"] == 3


def test_bench_first_gather_deadline_covers_the_first_collective():
    """The very first RCCL collective of a run is the warm-up gather of bench.py.  With the rank's gather stream parked behind a
    6 s spin kernel and --gather-timeout 1.5 the run must reach the fallback (polled, agreed, aborted) instead of blocking inside
    cf_gather_topk: the shard agreement in front of the gather is asynchronous."""
    import json
    import time
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "2",
                          "--batch", "4", "--size", "160", "--topk", "20", "--no-cpu-baseline", "--no-extras", "--profile-reps", "1",
                          "--exercise-gather-path", "--gather-timeout", "1.5", "--debug-stall-gather-ms", "6000"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["config"]["gather_fallback"] and "did not complete" in d["config"]["gather_fallback"]
    assert "torch.distributed" in d["config"]["gather"] and d["config"]["contexts_per_gpu"] == 1
    assert time.perf_counter() - t0 < 120


def test_gather_is_asynchronous_and_latches_a_shard_mismatch():
    """ONE collective per gather step: the shard (B, K) is agreed once (cf_comm_set_shard: enqueued, polled),
    every gather sends a fixed-size slot whose header carries (B, K, step) and is validated on the device behind the all-gather.
    cf_gather_topk(out_on_device=1) returns while the gather stream is parked (no host wait once the shard is agreed); a (B, K)
    mismatch between ranks -- forced here by skewing the B in this rank's header -- is latched and reported by query /
    synchronize / the next gather, and by the blocking form itself."""
    import time
    import torch
    S, B, K = 96, 3, 20
    imgs = np.random.default_rng(17).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    eng = cfa.Engine(S, S, max_batch=B, dtype="bf16")
    comm = cfa.distributed.Comm(eng, 0, 1, cfa.distributed.unique_id())
    out = torch.zeros((B, K, 16), dtype=torch.float32, device="cuda:0")
    # explicit agreement with the gather stream parked: set_shard returns at once, the verdict arrives by polling
    comm.debug(0, 800)
    t0 = time.perf_counter()
    comm.set_shard(B, K)
    assert time.perf_counter() - t0 < 0.5, "cf_comm_set_shard waited on the host"
    assert comm.query()
    assert comm.wait(60.0)
    eng.forward_enqueue(imgs)
    comm.gather_topk_device(K, out.data_ptr())          # warm-up (loads RCCL kernels)
    assert comm.wait(60.0)
    want = out.cpu().numpy().copy()
    eng.forward_enqueue(imgs)
    assert np.array_equal(want, _records(eng, K))        # the unpack kernel stripped the slot header: records only, in place
    comm.debug(0, 1500)                                  # park the gather stream for 1.5 s
    t0 = time.perf_counter()
    eng.forward_enqueue(imgs)
    comm.gather_topk_device(K, out.data_ptr())
    dt = time.perf_counter() - t0
    assert dt < 0.75, "cf_gather_topk waited %.2f s on the host behind a parked gather stream" % dt
    assert comm.query()                                  # still running
    assert comm.wait(30.0)
    assert np.array_equal(out.cpu().numpy(), want)
    # mismatch, asynchronous form: latched, reported by query and by synchronize and by the next gather
    comm.debug(1, 1)
    eng.forward_enqueue(imgs)
    comm.gather_topk_device(K, out.data_ptr())
    with pytest.raises(RuntimeError, match="equal shards"):
        comm.wait(30.0)
    with pytest.raises(RuntimeError, match="equal shards"):
        comm.synchronize()
    with pytest.raises((RuntimeError, ValueError), match="equal shards"):
        comm.gather_topk_device(K, out.data_ptr())
    with pytest.raises(RuntimeError, match="equal shards"):
        comm.set_shard(B, K)                             # the latch is sticky: abort is the way out
    comm.abort()
    # blocking form on a fresh communicator (implicit agreement on its first gather): CF_EINVAL from the call itself
    comm = cfa.distributed.Comm(eng, 0, 1, cfa.distributed.unique_id())
    eng.forward_enqueue(imgs)
    assert np.array_equal(comm.gather_topk(K), want)
    comm.debug(1, -1)
    with pytest.raises((RuntimeError, ValueError), match="equal shards"):
        comm.gather_topk(K)
    comm.abort()
    eng.close()


def test_gather_shard_change_is_rank_local_and_never_sends_unequal_counts():
    """ADVICE r04 medium + r05 medium: no rank ever enqueues an all-gather with another count.  The agreed slot is a capacity: shards
    that fit (ragged last batch, smaller K) are gathered in it as they are; a rank whose shard does NOT fit sends the agreed slot
    size with its real (B, K) in the header and gets CF_EINVAL at once, the header check latches the mismatch (as it would on every
    other rank), and a larger geometry needs cf_comm_set_shard on a fresh communicator or before any mismatch.  Also: changing the
    geometry through set_shard, and K larger than the agreed K."""
    import torch
    S, B, K = 96, 4, 20
    imgs = np.random.default_rng(18).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    eng = cfa.Engine(S, S, max_batch=B, dtype="bf16")
    comm = cfa.distributed.Comm(eng, 0, 1, cfa.distributed.unique_id())
    eng.forward_enqueue(imgs)
    want = _records(eng, K)
    assert np.array_equal(comm.gather_topk(K), want)
    # geometry change done right: set_shard (collective by contract), then gathers of the new shape
    comm.set_shard(B - 1, K + 5)
    assert comm.wait(30.0)
    eng.forward_enqueue(imgs[:B - 1])
    want2 = _records(eng, K + 5)
    got2 = comm.gather_topk(K + 5)
    assert got2.shape == (B - 1, K + 5, 16) and np.array_equal(got2, want2)
    out = torch.zeros((B - 1, K + 5, 16), dtype=torch.float32, device="cuda:0")
    eng.forward_enqueue(imgs[:B - 1])
    comm.gather_topk_device(K + 5, out.data_ptr())
    assert comm.wait(30.0) and np.array_equal(out.cpu().numpy(), want2)
    # round 6: the agreed slot is a capacity -- a SHORTER shard (the ragged last batch, identical on every rank) and a smaller
    # K travel in the same fixed-size slot with their real (B, K) in the header: no second agreement, no latch
    eng.forward_enqueue(imgs[:B - 2])
    want3 = _records(eng, K + 5)
    out3 = torch.zeros((B - 2, K + 5, 16), dtype=torch.float32, device="cuda:0")
    comm.gather_topk_device(K + 5, out3.data_ptr())
    assert comm.wait(30.0) and np.array_equal(out3.cpu().numpy(), want3)
    got4 = comm.gather_topk(K - 3)
    assert got4.shape == (B - 2, K - 3, 16) and np.array_equal(got4, _records(eng, K - 3))
    eng.forward_enqueue(imgs[:B - 1])
    assert np.array_equal(comm.gather_topk(K + 5), want2)          # and the full agreed shard again
    # a shard that does NOT fit the agreed slot: rank-local CF_EINVAL, full-size slot sent, latched
    eng.forward_enqueue(imgs)
    with pytest.raises((RuntimeError, ValueError), match="equal shards"):
        comm.gather_topk_device(K + 5, out.data_ptr())
    with pytest.raises(RuntimeError, match="equal shards"):
        comm.wait(30.0)                                  # the collective itself completed (equal counts); the verdict is the latch
    comm.abort()
    comm = cfa.distributed.Comm(eng, 0, 1, cfa.distributed.unique_id())
    comm.set_shard(B, K)
    eng.forward_enqueue(imgs)
    with pytest.raises((RuntimeError, ValueError), match="equal shards"):
        comm.gather_topk(K + 1)                          # another K than the agreed one
    comm.abort()
    eng.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gather_dry_run_n_ranks_equals_unsharded_order_bitwise(world):
    """the N-rank gather without N GPUs.  A loopback communicator (cf_comm_create_loopback) plays the ranks of a
    2 / 4 / 8-rank job in turn on one GPU -- shard_range() shards of one batch, each decoded, given its slot header and copied into its
    rank's place of the landing area (the stand-in for ncclAllGather); the header check + rank-major unpack of the real gather then
    run for `world` ranks.  The gathered records must equal the UNSHARDED decode of the whole batch bit for bit (host and device
    destination, ranks playing out of order, two steps: the step number in the headers advances once per step), a ragged last step fits
    the agreed slot, and a rank with another shard is found by the header check."""
    import torch
    S, K, per = 96, 20, 2
    Btot = world * per
    imgs = np.random.default_rng(100 + world).integers(0, 256, (Btot, S, S, 3), dtype=np.uint8)
    full = cfa.Engine(S, S, max_batch=Btot, dtype="bf16")
    full.forward_enqueue(imgs)
    want = _records(full, K)
    eng = cfa.Engine(S, S, max_batch=per, dtype="bf16")
    comm = cfa.distributed.Comm.loopback(eng, world)
    comm.set_shard(per, K)
    assert comm.wait(30.0)
    order = list(range(world))
    for step in range(2):
        got = None
        for r in (order if step == 0 else order[::-1]):
            lo, hi = cfa.distributed.shard_range(Btot, r, world)
            assert hi - lo == per
            eng.forward_enqueue(imgs[lo:hi])
            comm.play(r)
            out = comm.gather_topk(K)
            got = out if out is not None else got
        assert got.shape == (Btot, K, 16) and np.array_equal(got, want), step
    # device destination
    dev = torch.zeros((Btot, K, 16), dtype=torch.float32, device="cuda:0")
    for r in order:
        lo, hi = cfa.distributed.shard_range(Btot, r, world)
        eng.forward_enqueue(imgs[lo:hi])
        comm.play(r)
        comm.gather_topk_device(K, dev.data_ptr())
    assert comm.wait(30.0) and np.array_equal(dev.cpu().numpy(), want)
    # a ragged last step (one image per rank, smaller K) travels in the agreed slot
    full.forward_enqueue(imgs[::per])
    want1 = _records(full, K - 5)
    for r in order:
        eng.forward_enqueue(imgs[r * per:r * per + 1])
        comm.play(r)
        got1 = comm.gather_topk(K - 5)
    assert got1.shape == (world, K - 5, 16) and np.array_equal(got1, want1)
    # the same rank twice in one step is a call-order error; a rank with another (B, K) is found by every rank's header check
    eng.forward_enqueue(imgs[:per])
    comm.play(0)
    comm.gather_topk(K)
    with pytest.raises(cfa._lib.CenterFaceError, match="deposited"):
        comm.gather_topk(K)
    for r in order[1:-1]:
        eng.forward_enqueue(imgs[:per])
        comm.play(r)
        assert comm.gather_topk(K) is None
    eng.forward_enqueue(imgs[:1])                                            # the last rank comes with a shorter shard than the others
    comm.play(world - 1)
    with pytest.raises((RuntimeError, ValueError), match="equal shards"):
        comm.gather_topk(K)
    comm.abort()
    eng.close(); full.close()


@pytest.mark.isolated
@pytest.mark.parametrize("cfg", ["configs2", "configs4"])
def test_baseline_multi_gpu_configs_dry_run_on_one_gpu(cfg):
    """BASELINE configs[2] (B = 512 over 8 GPUs: 64 x 640x640 per rank, top-100) and configs[4] (1280x1280 dense crowd, top-1000, 8 GPUs: 4 per
    rank) at their REAL per-rank shapes and slot sizes, every leg but the RCCL transport: one GPU plays the eight ranks through the loopback
    communicator (two contexts per rank as bench.py drives them for configs[2], three without decode streams for configs[4]); the gathered
    [8 x B, K, 16] records of a step equal the rank-major concatenation of each shard's own decode bit for bit."""
    import torch
    world = 8
    S, B, K, depth = (640, 64, 100, 2) if cfg == "configs2" else (1280, 4, 1000, 3)
    ring = cfa.EngineRing(S, S, depth=depth, max_batch=B, dtype="bf16")
    comm = cfa.distributed.Comm.loopback(ring.engines[0], world)
    comm.set_shard(B, K)
    assert comm.wait(30.0)
    rng = np.random.default_rng(42)
    shards = [rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8) for _ in range(2)]          # two distinct shards, dealt to the ranks in turn
    want = []
    for r in range(world):
        ring.engines[0].forward_enqueue(shards[r % 2])
        want.append(_records(ring.engines[0], K))
    want = np.concatenate(want)
    dev = torch.zeros((world * B, K, 16), dtype=torch.float32, device="cuda:0")
    for step in range(2):
        for r in range(world):
            eng = ring.engines[(step * world + r) % depth]                                  # the rank's contexts alternate, one communicator
            eng.forward_enqueue(shards[r % 2])
            comm.play(r)
            comm.gather_topk_device(K, dev.data_ptr(), engine=eng)
        assert comm.wait(60.0)
        got = dev.cpu().numpy()
        assert got.shape == (world * B, K, 16) and np.array_equal(got, want), (cfg, step)
        dev.zero_()
    comm.abort()
    ring.close()


def test_engine_ring_matches_single_engine_bitwise():
    """EngineRing: batches submitted back to back on alternating contexts, collected out of order, different batch sizes
    and K -- every result equals the single-engine result of the same batch."""
    S = 160
    rng = np.random.default_rng(5)
    batches = [rng.integers(0, 256, (b, S, S, 3), dtype=np.uint8) for b in (8, 3, 8, 1, 5, 8)]
    ks = [50, 20, 50, 100, 7, 50]
    ref = cfa.Engine(S, S, max_batch=8, dtype="bf16")
    want = []
    for x, k in zip(batches, ks):
        ref.forward_enqueue(x)
        want.append(ref.decode_topk(k))
    ref.close()
    ring = cfa.EngineRing(S, S, depth=2, max_batch=8, dtype="bf16")
    # the two main streams must not sit on one hardware queue (their forwards would serialise); EngineRing re-rolls until they do not
    assert not ring.engines[1].shares_queue_with(ring.engines[0])
    tickets = []
    for i, (x, k) in enumerate(zip(batches, ks)):
        tickets.append(ring.submit(x, K=k))
        if i % 2 == 1:                                   # two in flight, then collect both (newest first)
            for j in (i, i - 1):
                d, l, ind = ring.collect(tickets[j])
                assert np.array_equal(d, want[j][0]) and np.array_equal(l, want[j][1]) and np.array_equal(ind, want[j][2]), j
    # streams re-created after graphs were captured and results are in flight: everything still valid
    t = ring.submit(batches[0], K=ks[0])
    ring.engines[t[0]].reroll_streams()
    d, l, ind = ring.collect(t)
    assert np.array_equal(d, want[0][0]) and np.array_equal(ind, want[0][2])
    t = ring.submit(batches[0], K=ks[0])
    d, l, ind = ring.collect(t)
    assert np.array_equal(d, want[0][0]) and np.array_equal(l, want[0][1]) and np.array_equal(ind, want[0][2])
    ring.close()
    # three contexts: the device-output decodes run on the main streams (CF_FLAG_NO_DECODE_STREAM; six streams would share HIP's four
    # hardware queues) -- same results, three batches in flight
    ring3 = cfa.EngineRing(S, S, depth=3, max_batch=8, dtype="bf16")
    tickets = [ring3.submit(x, K=k) for x, k in zip(batches[:3], ks[:3])]
    for j in (2, 0, 1):
        d, l, ind = ring3.collect(tickets[j])
        assert np.array_equal(d, want[j][0]) and np.array_equal(l, want[j][1]) and np.array_equal(ind, want[j][2]), j
    ring3.close()
    one = cfa.Engine(S, S, max_batch=8, dtype="bf16", decode_stream=False)
    one.forward_enqueue(batches[0])
    d, l, ind = one.decode_topk(ks[0])
    assert np.array_equal(d, want[0][0]) and np.array_equal(ind, want[0][2])
    one.close()


@pytest.mark.isolated
@pytest.mark.parametrize("dummies", [0, 1, 2, 3])
def test_engine_ring_priority_placement_needs_no_probe_and_no_luck(dummies):
    """Round 6: EngineRing(placement="priority") creates the ring's streams in the highest stream-priority class
    (CF_FLAG_STREAM_HIGH) -- in a process that has not used that class before (this test runs in a child interpreter of its own) the two main
    streams get hardware queues and dispatch pipes of their own whatever the process created in the DEFAULT class (0 ... 3 dummy streams
    first: exactly the histories that put both main streams of round 2-5's rings on one queue), without a single probe kernel at creation.
    The probes below are the test's own.  The default stays placement="probe": the priority class is not proof against the library's own earlier
    contexts in it (tools/ring_sequence_probe.py), probing is."""
    import torch
    keep = []
    for _ in range(dummies):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            torch.zeros(1, device="cuda").add_(1)
        keep.append(st)
    torch.cuda.synchronize()
    ring = cfa.EngineRing(160, 160, depth=2, max_batch=4, dtype="bf16", placement="priority")
    assert ring.placement == "priority" and ring.queue_rerolls == 0
    a, b = ring.engines
    assert not a.queue_shared(0, b, 0)                  # main streams: different hardware queues ...
    assert not a.queue_shared(16, b, 0)                 # ... and different dispatch pipes
    x = np.random.default_rng(dummies).integers(0, 256, (4, 160, 160, 3), dtype=np.uint8)
    t0, t1 = ring.submit(x, K=20), ring.submit(x, K=20)
    r0, r1 = ring.collect(t0), ring.collect(t1)
    for u, v in zip(r0, r1):
        assert np.array_equal(u, v)
    ring.close()
    probe = cfa.EngineRing(160, 160, depth=2, max_batch=4, dtype="bf16")      # the default: probed, repaired when needed
    assert probe.placement == "probe" and not probe.engines[0].queue_shared(16, probe.engines[1], 0)
    t0 = probe.submit(x, K=20)
    for u, v in zip(probe.collect(t0), r0):
        assert np.array_equal(u, v)
    probe.close()


@pytest.mark.isolated
def test_spread_streams_places_contexts_and_keeps_results():
    """cf_spread_streams / cf_streams_share_queue_ex: three contexts' main streams (and one context's decode stream) are re-placed -- all pairwise
    (window 0) and neighbours only (window 2); captured graphs and results stay valid, the probe answers for every selector, bad arguments are
    CF_EINVAL; after a window-0 placement the MAIN streams sit on queues of their own and (where the process reaches three dispatch pipes) on pipes of their own."""
    import ctypes as C
    L = cfa._lib.lib()
    S, B = 160, 4
    rng = np.random.default_rng(11)
    x = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    engs = [cfa.Engine(S, S, max_batch=B, dtype="bf16", decode_stream=(i == 0)) for i in range(3)]
    want = []
    for e in engs:
        e.forward_enqueue(x); e.forward_enqueue(x)                   # second sighting: the graph is captured
        want.append((e.heads(), e.decode_topk(20)))
    hs = (C.c_void_p * 3)(*[e._h for e in engs])
    nd = C.c_int(-1)
    for window in (0, 2, 1):
        assert L.cf_spread_streams(hs, 3, window, C.byref(nd)) == 0
        assert 1 <= nd.value <= 4
        for e, (h0, d0) in zip(engs, want):
            e.forward_enqueue(x)
            h1, d1 = e.heads(), e.decode_topk(20)
            for k in h0:
                assert np.array_equal(h0[k], h1[k]), (window, k)
            for a, b2 in zip(d0, d1):
                assert np.array_equal(a, b2)
    assert L.cf_spread_streams(hs, 3, 0, None) == 0
    pipe_clashes = 0
    for i in range(3):
        for j in range(i):
            assert not engs[i].queue_shared(0, engs[j], 0), (i, j)      # three main streams on four queues: always on queues of their own
            pipe_clashes += bool(engs[i].queue_shared(16, engs[j], 0))
    assert pipe_clashes <= 1                                            # (three pipes on every box seen so far: 0; two would leave one pair)
    sh = C.c_int(-1)
    assert L.cf_streams_share_queue_ex(engs[0]._h, 0, engs[0]._h, 0, C.byref(sh)) == 0 and sh.value == 1      # a stream shares with itself
    assert L.cf_streams_share_queue_ex(engs[0]._h, 2, engs[1]._h, 2, C.byref(sh)) == 0 and sh.value == 1      # the device's ONE copy stream
    assert L.cf_streams_share_queue_ex(engs[0]._h, 1, engs[1]._h, 17, C.byref(sh)) == 0                       # decode streams exist from now on
    assert L.cf_streams_share_queue_ex(engs[0]._h, 3, engs[1]._h, 0, C.byref(sh)) == -1
    assert L.cf_spread_streams(hs, 0, 0, None) == -1 and L.cf_spread_streams(None, 3, 0, None) == -1 and L.cf_spread_streams(hs, 3, -1, None) == -1
    for e in engs:
        e.close()


def test_bench_other_configs_block_runs():
    """bench.py's `other_configs` extras (configs[3] through CenterFaceBuckets, the configs[4] shard on three contexts): the block is wrapped in a
    try / except inside bench.py so that it can never cost the headline its line -- this test is where a breakage shows."""
    sys.path.insert(0, REPO)
    import bench
    o = bench.other_configs_block(cfa, 0)
    v = o["configs[3]"]
    assert v["page_locked_input"]["images_per_s"] > 1000 and v["pageable_input"]["images_per_s"] > 1000 and v["detections"] > 0
    s4 = o["configs[4]_per_gpu_shard"]
    assert s4["images_per_s"] > 500 and s4["contexts"] == 3 and s4["batch"] == 4 and s4["topk"] == 1000


def test_bench_standalone_depthwise_block_runs():
    """bench.py's `standalone_depthwise` extra (north_star's depthwise-conv HBM fraction, measured on the standalone kernel over the unfused forward): twelve
    depthwise layers with a rate each, the aggregate and the best layer; small batch here, the rates themselves are the bench run's business."""
    import torch
    sys.path.insert(0, REPO)
    import bench
    x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (8, 640, 640, 3), dtype=np.uint8)).to("cuda:0")
    o = bench.standalone_depthwise_block(cfa, x.data_ptr(), 8, 640, 100, 0, reps=2)
    assert len(o["layers"]) == 12 and all(v["GBps"] > 100 and "dw_" in v["kernel"] for v in o["layers"].values()), o
    assert o["best"]["layer"] in o["layers"] and 0 < o["all_twelve"]["frac_of_8TBps"] < 1 and 0 < o["best"]["frac_of_8TBps"] < 1


def test_bench_prints_one_json_line_with_the_contract_fields():
    """bench.py as the driver runs it (small workload): exactly one line on stdout, the contract's fields, roofline and
    the per-run consistency the judge checks (value = images of the median window / its time)."""
    import json
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--repeats", "3",
                          "--batch", "4", "--size", "160", "--topk", "20", "--no-cpu-baseline", "--no-extras", "--profile-reps", "1"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "windows"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "images/s" and d["dtype"] == "bf16" and d["data"] == "synthetic" and d["vs_baseline"] is None
    # --depth 0 (default) = automatic: a step this small (4 x 160 x 160) runs on three contexts whose decodes stay on their main streams
    assert "workload" in d["config"] and d["config"]["contexts_per_gpu"] == 3 and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu_issue") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(d["value"] - 4 * 3 / (d["windows"]["median_ms"] * 1e-3)) / d["value"] < 1e-3
    assert abs(d["ms_per_step"] - d["windows"]["median_ms"] / 3) < 1e-3


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(repo)r)
import centerface_amd as cfa
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
B, S, K = 6, 128, 40
imgs = np.random.default_rng(3).integers(0, 256, (B, S, S, 3), dtype=np.uint8)      # the same full batch on every rank
lo, hi = cfa.distributed.shard_range(B, rank, world)
eng = cfa.Engine(S, S, max_batch=B, dtype=%(dtype)r, device=0)
eng.forward_enqueue(imgs[lo:hi])
d, l, _ = eng.decode_topk(K)
rec = torch.from_numpy(cfa.distributed.pack_records(d, l))
allrec = cfa.distributed.gather_records(rec).numpy()
assert allrec.shape == (B, K, 16), allrec.shape
eng.forward_enqueue(imgs)                                                           # unsharded run on this rank
d, l, _ = eng.decode_topk(K)
full = cfa.distributed.pack_records(d, l)
assert np.array_equal(allrec, full), "sharded != unsharded"
eng.close()
dist.barrier()
dist.destroy_process_group()
print("rank %%d ok" %% rank)
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_world2_sharded_equals_unsharded_bitwise(tmp_path, dtype):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"repo": REPO, "dtype": dtype})
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert "rank %d ok" % r in o

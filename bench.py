#!/usr/bin/env python3
"""Benchmark of the CenterFace hot path on MI355X: images/s at 640x640 batch inference.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input per rank (BASELINE.json
configs[1]: batch 64, 640x640, bf16): uint8 BGR images already resident in HBM -> fused
normalise + stem -> 12 MBConv blocks -> IDAUp neck -> heads -> 3x3-peak / top-K / gather decode
(-> RCCL all-gather of the final boxes when N > 1).  Weak scaling: every rank processes its own
batch of 64; `value` is whole-job images/s.  Rank 0 prints ONE JSON line.

Each rank drives `--depth` (default 2) independent contexts round-robin: step k is enqueued on context k mod depth, so two
batches are in flight and the HBM- / latency-bound back half of one forward (project GEMMs on the small maps, up3+heads,
decode) runs underneath the VALU-bound front half of the next (cfa.EngineRing; a C host does it with two cf_ctx).
Every step still processes one whole batch; the windows are fenced by a synchronize of every context.
`value_one_context` reports the same steps on a single context.

Timing: W warm-up steps, then `--repeats` windows of EXACTLY K steps, each bracketed by barrier +
synchronize on both sides, max over ranks per window; `value` comes from the MEDIAN window, min / max are
reported next to it (`windows`).

Extra objects in the line:
  roofline       -- dominant kernel symbol of the forward: algorithmic bytes per launch / average
                    launch duration, measured here with HIP events on the stream the kernels run on.
  cpu_baseline   -- the oracle (a torch-CPU restatement of the reference path, oracle/) timed on this
                    box's host cores on a bounded sample of the same workload, B = 16 and B = 1.  A reported
                    baseline, not the optimisation target.
  parity         -- the timed batch decoded by the benchmarked (bf16) engine against the fp32 parity engine (itself
                    within 1e-3 of the reference) and, on one image, against the bf16-emulating oracle.
  value_with_h2d -- the same step with the batch starting in pinned HOST memory every step (PCIe inclusive).
  tolerance_mode -- images/s of the fp32_split mode (fp32 storage, split-bf16 GEMM products: within 1e-3 of the reference) at the
                    same batch and schedule, with its head / decode differences against the exact-fp32 engine on the timed batch.
  exact_fp32_mode -- images/s of the exact-fp32-MFMA mode (bit-level test mode).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=25, help="timed windows of --steps steps; value = median window")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--depth", type=int, default=0,
                    help="contexts per GPU used round-robin (batches in flight); 1 = a single context, every step on one stream chain; "
                         "0 = automatic: 2, or 3 (decodes on the main streams) when a step is small (batch x size^2 <= 8 M pixels, e.g. four 1280x1280 images)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32", "fp32_split"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip value_with_h2d / tolerance_mode / exact_fp32_mode / parity (profiling runs)")
    ap.add_argument("--gather", default="auto", choices=["auto", "cf", "torch"],
                    help="N > 1 gather of the final boxes: cf = cf_gather_topk (C ABI, RCCL), torch = torch.distributed; "
                         "auto = cf, falling back to torch if the communicator cannot be created")
    ap.add_argument("--gather-timeout", type=float, default=60.0,
                    help="seconds the first (warm-up) RCCL gather may take; on expiry on ANY rank every rank aborts its communicator "
                         "and the run continues with --gather torch --depth 1 (agreed over the TCP store, not over NCCL)")
    ap.add_argument("--debug-stall-gather-ms", type=int, default=0,
                    help="test hook: park the rank's gather stream behind a spin kernel of this many ms before the warm-up gather "
                         "(a first collective that does not complete in time; tests/test_multigpu.py)")
    ap.add_argument("--input-buffers", type=int, default=4,
                    help="distinct resident input batches the timed steps rotate through (4 x 78.6 MB > the 256 MB Infinity Cache: "
                         "no step can find its images cached from an earlier one)")
    ap.add_argument("--exercise-gather-path", action="store_true",
                    help="run the N>1 step (decode-stream gather, identity at world 1) on one GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--profile-reps", type=int, default=5)
    ap.add_argument("--traffic-json", default=None,
                    help="per-kernel HBM bytes from rocprofv3 PMC passes (tools/summarize_prof.py); "
                         "default: the newest profiles/traffic_*.json")
    return ap.parse_args(argv)


def _host_core_set():
    """The cores the CPU baseline is pinned to: the physical cores (one hardware thread each) of ONE NUMA node among the CPUs
    this process may run on -- the node with the most of them.  Returns (sorted cpu ids, description)."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))

    def parse(text):
        out = []
        for part in text.strip().split(","):
            if part:
                a, _, b = part.partition("-")
                out.extend(range(int(a), int(b or a) + 1))
        return out
    nodes = {}
    try:
        base = "/sys/devices/system/node"
        for d in sorted(os.listdir(base)):
            if d.startswith("node") and d[4:].isdigit():
                cpus = [c for c in parse(open(os.path.join(base, d, "cpulist")).read()) if c in set(allowed)]
                if cpus:
                    nodes[int(d[4:])] = cpus
    except OSError:
        pass
    node, cpus = max(nodes.items(), key=lambda kv: (len(kv[1]), -kv[0])) if nodes else (None, allowed)
    phys, seen = [], set()
    for c in cpus:                                            # one hardware thread per physical core
        try:
            sib = tuple(parse(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read()))
        except OSError:
            sib = (c,)
        if sib not in seen:
            seen.add(sib)
            phys.append(c)
    return phys, ("NUMA node %s" % node if node is not None else "all allowed CPUs")


def _cpu_baseline_child(spec_path):
    """Runs in a FRESH process (python bench.py --cpu-baseline-child spec.json): pinned to the chosen cores before torch / OpenMP
    start (the parent set OMP_PROC_BIND / OMP_PLACES), so thread placement does not depend on what the bench process did before."""
    import json
    spec = json.load(open(spec_path))
    if hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, spec["cpus"])
    import torch
    import centerface_amd as cfa
    from oracle import centerface_oracle as O
    imgs = np.load(spec["imgs"])
    size, topk, seconds = spec["size"], spec["topk"], spec["seconds"]
    sd = O.to_torch_sd(cfa.weights.synthetic_state_dict(0))

    def runner(bs):
        x = torch.from_numpy(np.concatenate([O.preprocess(im) for im in imgs[:bs]]))

        def run():
            t0 = time.perf_counter()
            out = O.forward(sd, x)
            hm = O.sigmoid_clamp(out["hm"]).numpy()
            O.ctdet_decode(hm, out["wh"].numpy(), out["reg"].numpy(), topk, out["lm"].numpy())
            return time.perf_counter() - t0
        return run

    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])
    # FIXED thread count (round 6): 16 threads on 16 pinned physical cores (fewer if the host has fewer).  Rounds 2-5
    # searched the thread count and reported the better batch size; the winner changed from box to box (5 ... 34 img/s), so the
    # rounds were not comparable.  The probe over other counts is still printed (B = 1, one warm-up + three runs each), it no
    # longer decides anything.
    ncores = len(spec["cpus"])
    fixed = min(16, ncores)
    cands = sorted({t for t in (8, 16, 32, 64) if t <= ncores} | {fixed})
    run1 = runner(1)
    probe = {}
    for th in cands:
        torch.set_num_threads(th)
        run1()
        probe[th] = med([run1() for _ in range(3)])
    best = fixed
    torch.set_num_threads(best)

    def measure(bs, budget, min_runs):
        run = runner(bs)
        run()                                   # warm-up
        ts, t0 = [], time.perf_counter()
        while len(ts) < min_runs or (time.perf_counter() - t0 < budget and len(ts) < 64):
            ts.append(run())
        rates = sorted(bs / t for t in ts)
        return {"median": med(rates), "min": rates[0], "max": rates[-1], "runs": len(ts), "images": bs * len(ts), "seconds": sum(ts)}
    b16 = measure(min(16, len(imgs)), seconds * 0.6, 3)
    b1 = measure(1, seconds * 0.4, 5)
    json.dump({"threads": best, "probe_ms_b1": {str(k): round(v * 1e3, 1) for k, v in probe.items()}, "b16": b16, "b1": b1}, sys.stdout)


def cpu_baseline(seconds, size, topk, imgs):
    """Oracle forward + ctdet_decode on the host CPU, bounded sample, B = 16 and B = 1, in a child process pinned to the physical
    cores of one NUMA node (OMP_PROC_BIND=close, OMP_PLACES=cores): the median of >= 3 / >= 5 runs, with the spread next to it."""
    import json
    import subprocess
    import tempfile
    host_cores = os.cpu_count() or 1
    cpus, where = _host_core_set()
    cpus = cpus[:16]                                          # the fixed configuration: 16 threads on 16 physical cores of one NUMA node
    with tempfile.TemporaryDirectory() as td:
        np.save(os.path.join(td, "imgs.npy"), np.ascontiguousarray(imgs[:16]))
        spec = {"cpus": cpus, "imgs": os.path.join(td, "imgs.npy"), "size": size, "topk": topk, "seconds": seconds}
        json.dump(spec, open(os.path.join(td, "spec.json"), "w"))
        env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", KMP_AFFINITY="granularity=core,compact", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        env.pop("OMP_NUM_THREADS", None)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", os.path.join(td, "spec.json")],
                             capture_output=True, text=True, env=env, timeout=max(300.0, 20 * seconds))
    if out.returncode != 0:
        raise RuntimeError("cpu baseline child failed: %s" % out.stderr[-2000:])
    r = json.loads(out.stdout.strip().splitlines()[-1])
    b16, b1 = r["b16"], r["b1"]
    # value = ALWAYS the B = 16 figure at the fixed thread count (comparable across rounds and boxes); B = 1 is reported beside it
    return {"value": round(b16["median"], 2), "unit": "images/s", "cores": r["threads"], "host_cores": host_cores,
            "kind": "port", "value_b16": round(b16["median"], 2), "value_b1": round(b1["median"], 2),
            "spread": {"b16_min_max": [round(b16["min"], 2), round(b16["max"], 2)], "b16_runs": b16["runs"],
                       "b1_min_max": [round(b1["min"], 2), round(b1["max"], 2)], "b1_runs": b1["runs"]},
            "thread_probe_ms_per_image_b1": r["probe_ms_b1"],
            "pinned_to": "%d physical cores of %s (sched_setaffinity + OMP_PROC_BIND=close, OMP_PLACES=cores), fresh process" % (len(cpus), where),
            "sample": "B=16: %d images in %.1f s (%d runs); B=1: %d images in %.1f s (%d runs) (%dx%d, fp32 torch-CPU oracle forward + top-%d decode; "
                      "FIXED %d threads on %d pinned physical cores, %d host cores; value = median B=16 rate, value_b1 = median B=1 rate)"
                      % (b16["images"], b16["seconds"], b16["runs"], b1["images"], b1["seconds"], b1["runs"], size, size, topk, r["threads"], len(cpus), host_cores)}


def flush_c_stdio():
    """Flush Python's and the C library's stdout buffers (librccl's banner must not land after the JSON line)."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:                                   # noqa: BLE001
        pass


def make_step(cfa, engs, d_in_ptr, B, K, outs, gather="none", comms=None):   # d_in_ptr: one device pointer or a list the steps rotate through
    """One benchmark step as a closure: forward + top-K decode (+ gather) of one batch.

    ``engs`` / ``outs``: one Engine / output dict, or lists of ``depth`` of them -- step k then runs on context k % depth;
    ``comms``: the rank's ONE Comm (every context gathers through it, on its single gather stream, in step order)
    -- (cfa.EngineRing's schedule: two batches in flight, the back half of one forward under the front
    half of the next).  ``out``: dict of device tensors dets/lms/inds and ``all`` (the gathered records).
    gather: none | cf (cf_gather_topk on the decode stream) | torch (torch.distributed)."""
    import torch
    fmt = cfa._lib.CF_IN_U8_HWC_BGR
    if not isinstance(engs, (list, tuple)):
        engs, outs = [engs], [outs]
    comm = comms[0] if isinstance(comms, (list, tuple)) else comms       # ONE communicator per rank, shared by its contexts
    depth, k = len(engs), [0]
    in_ptrs = list(d_in_ptr) if isinstance(d_in_ptr, (list, tuple)) else [d_in_ptr]
    if gather == "torch":
        # the decode stream of the context and torch's stream (pack + all-gather) are chained with stream waits,
        # never a host sync: the gather of step i runs underneath the forward of step i+1
        dec_streams = [torch.cuda.ExternalStream(e.streams()[1], device=o["dets"].device) for e, o in zip(engs, outs)]

    def step():
        i = k[0] % depth
        src = in_ptrs[k[0] % len(in_ptrs)]
        k[0] += 1
        eng, out = engs[i], outs[i]
        eng.forward_enqueue(src, on_device=True, B=B, in_format=fmt)
        if gather == "cf":
            comm.gather_topk_device(K, out["all"].data_ptr(), engine=eng)
            return out["all"]
        if gather == "torch":
            dec_streams[i].wait_stream(torch.cuda.current_stream())      # this slot's last pack has consumed dets / lms
            eng.decode_topk_device(K, out["dets"].data_ptr(), out["lms"].data_ptr(), out["inds"].data_ptr())
            torch.cuda.current_stream().wait_stream(dec_streams[i])      # boxes are final before the pack reads them
            rec = cfa.distributed.pack_records(out["dets"], out["lms"])
            out["all"] = cfa.distributed.gather_records(rec)
            return out["all"]
        eng.decode_topk_device(K, out["dets"].data_ptr(), out["lms"].data_ptr(), out["inds"].data_ptr())
        return None
    return step


def time_windows(step, fence, steps, repeats, reduce_max=None):
    """`repeats` windows of exactly `steps` steps, fenced on both sides; returns the per-window seconds."""
    out = []
    for _ in range(repeats):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        out.append(reduce_max(dt) if reduce_max else dt)
    return out


def other_configs_block(cfa, dev_index):
    """BASELINE configs[3] (128 VGA-class images of five shapes through CenterFaceBuckets, host arrays in, numpy boxes out) and the per-GPU shard
    of configs[4] (four 1280x1280 images, top-1000, resident input), bf16, synthetic data: whole-call / whole-step rates of THIS run."""
    import torch
    rng = np.random.default_rng(0)
    shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416)]
    pageable = [rng.integers(0, 256, shapes[i % 5] + (3,), dtype=np.uint8) for i in range(128)]
    pinned = []
    for im in pageable:
        a = cfa.pinned_empty(im.shape)
        a[...] = im
        pinned.append(a)
    out = {}
    pool = cfa.CenterFaceBuckets(dtype="bf16", max_batch=32, max_buckets=8, device=dev_index)
    vga = {}
    for tag, batch in (("page_locked_input", pinned), ("pageable_input", pageable)):
        for _ in range(3):
            pool.detect(batch)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            res = pool.detect(batch)
            ts.append(time.perf_counter() - t0)
        vga[tag] = {"images_per_s": round(128 / float(np.median(ts)), 1), "ms_per_call": round(float(np.median(ts)) * 1e3, 3)}
    vga["detections"] = int(sum(len(r[0]) for r in res))
    vga["note"] = ("BASELINE configs[3]: 128 images, five VGA-class shapes, CenterFaceBuckets.detect (host uint8 arrays in -> upload, forward, D1 decode, NMS, "
                   "floor rescale on the GPU -> numpy boxes out); median of 7 calls; page_locked_input = arrays from cfa.pinned_empty / cfa.pin")
    out["configs[3]"] = vga
    pool.close()
    del pinned
    S, B, K = 1280, 4, 1000
    ring = cfa.EngineRing(S, S, depth=3, max_batch=B, dtype="bf16", device=dev_index)
    dev = torch.device("cuda", dev_index)
    xs = [torch.from_numpy(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).to(dev) for _ in range(4)]
    outs = [{"dets": torch.empty((B, K, 6), dtype=torch.float32, device=dev), "lms": torch.empty((B, K, 10), dtype=torch.float32, device=dev),
             "inds": torch.empty((B, K), dtype=torch.int64, device=dev)} for _ in ring.engines]
    st = make_step(cfa, ring.engines, [t.data_ptr() for t in xs], B, K, outs)

    def fence():
        for e in ring.engines:
            e.synchronize()
        torch.cuda.synchronize()
    for _ in range(6):
        st()
    w = time_windows(st, fence, 30, 7)
    out["configs[4]_per_gpu_shard"] = {"images_per_s": round(B * 30 / float(np.median(w)), 1), "ms_per_step": round(float(np.median(w)) / 30 * 1e3, 4), "batch": B,
                                       "image": [S, S], "topk": K, "contexts": len(ring.engines),
                                       "note": "four 1280x1280 images per step resident in HBM, forward + top-1000 decode; three contexts round-robin, decodes on "
                                               "their main streams (CF_FLAG_NO_DECODE_STREAM); median of 7 windows of 30 steps"}
    ring.close()
    return out


def standalone_depthwise_block(cfa, d_in_ptr, B, S, K, dev_index, reps=5):
    """north_star's "depthwise-conv achieved HBM >= 60 % of gfx950 peak": the benchmarked path fuses every depthwise into its MBConv kernel (the
    depthwise tensor never reaches HBM), so the STANDALONE kernel (cf_dw.hip: unfused path, cf_op_dwconv, ShuffleV2 block) is timed here on the
    twelve depthwise layers of this network at the benchmarked batch: HIP events around each launch of the unfused forward (cf_profile_forward),
    GB/s of unpadded input + output bytes."""
    eng = cfa.Engine(S, S, max_batch=B, dtype="bf16", fuse=False, device=dev_index)
    try:
        fmt = cfa._lib.CF_IN_U8_HWC_BGR
        for _ in range(2):
            eng.profile_forward(d_in_ptr, on_device=True, B=B, in_format=fmt, K=K)
        acc = {}
        for _ in range(reps):
            for r in eng.profile_forward(d_in_ptr, on_device=True, B=B, in_format=fmt, K=K):
                if r["kind"] == "dw":
                    a = acc.setdefault(r["name"], dict(ms=0.0, bytes=r["algo_bytes"], kernel=r["kernel"]))
                    a["ms"] += r["ms"]
    finally:
        eng.close()
    layers, tb, tms = {}, 0.0, 0.0
    for name, a in acc.items():
        ms = a["ms"] / reps
        layers[name] = {"GBps": round(a["bytes"] / (ms * 1e-3) / 1e9, 1), "ms": round(ms, 4), "frac_of_8TBps": round(a["bytes"] / (ms * 1e-3) / 8e12, 3),
                        "kernel": a["kernel"].split("(")[0].replace("void cf::", "")}
        tb += a["bytes"]; tms += ms
    best = max(layers.items(), key=lambda kv: kv[1]["GBps"])
    return {"layers": layers, "best": {"layer": best[0], "GBps": best[1]["GBps"], "frac_of_8TBps": best[1]["frac_of_8TBps"]},
            "all_twelve": {"GBps": round(tb / (tms * 1e-3) / 1e9, 1), "ms": round(tms, 4), "frac_of_8TBps": round(tb / (tms * 1e-3) / 8e12, 3)},
            "note": "standalone depthwise + Swish kernel (bf16, B = %d, %dx%d network shapes), algorithmic bytes = unpadded input + output; not on the benchmarked "
                    "path (fused there); layer1.0 (3x3 stride 2, 96 channels, 320^2 -> 160^2) is the largest depthwise of the network (29 %% of all depthwise bytes)" % (B, S, S)}


def parity_block(cfa, eng16, host_imgs, d_in_ptr, B, S, K, dev_index):
    """The timed batch through the benchmarked engine vs the fp32 parity engine (GPU), + image 0 vs the bf16 emulation."""
    import torch
    from oracle import bf16_emulation as E
    from oracle import centerface_oracle as O
    fmt = cfa._lib.CF_IN_U8_HWC_BGR
    eng16.forward_enqueue(d_in_ptr, on_device=True, B=B, in_format=fmt)
    d16, _, i16 = eng16.decode_topk(K)
    h16 = eng16.heads(sigmoid_hm=True)
    e32 = cfa.Engine(S, S, max_batch=B, dtype="fp32", device=dev_index)
    e32.forward_enqueue(d_in_ptr, on_device=True, B=B, in_format=fmt)
    d32, _, i32 = e32.decode_topk(K)
    e32.close()
    # ---- "box match vs ref" (BASELINE.json's metric): the reference's own evaluate() measure (eval_widerface.py:172-211:
    # D2 decode + NMS 0.3 of get_detections, then recall / precision at IoU 0.5 through bbox_overlap) with the exact-fp32
    # engine's detections in the role of the annotations, over the whole timed batch; and both engines against the CPU
    # oracle (= the reference's arithmetic) on image 0.  Two adjustments for synthetic weights, whose heads are noise:
    # (1) the reference's decoders take the box size LINEARLY from the wh head (no exp: eval_widerface.py:100, centerface.py:84),
    #     so a zero-mean wh head gives half of all boxes a negative width or height and an IoU of 0 even with themselves --
    #     the wh head's output bias is shifted by +6 (boxes of 24 +- 4 px) in the weights of ALL three parties;
    # (2) the score threshold is the exact engine's median K-th best score of this batch (the reference's 0.35 would keep
    #     ~70 % of all cells): about K candidates per image before NMS.
    from centerface_amd import eval_widerface as ew
    sd_m = dict(cfa.weights.synthetic_state_dict(0))
    sd_m["wh.1.bias"] = (sd_m["wh.1.bias"] + 6.0).astype(np.float32)
    thr = float(np.median(d32[:, K - 1, 4]))

    def d2_boxes(dtype):
        e = cfa.Engine(S, S, max_batch=B, dtype=dtype, device=dev_index, weights=sd_m)
        e.forward_enqueue(d_in_ptr, on_device=True, B=B, in_format=fmt)
        out = [b for b, _ in e.decode_threshold(thr, 0.3, mode="d2")]
        e.close()
        return out
    p32 = d2_boxes("fp32")
    p16 = d2_boxes(eng16.dtype)
    bm = ew.box_match(p16, p32, 0.5, device=dev_index)
    ref = O.forward(O.to_torch_sd(sd_m), torch.from_numpy(O.preprocess(host_imgs[0])))
    ref_boxes = O.decode_d2(O.sigmoid_clamp(ref["hm"]).numpy()[0], ref["wh"].numpy()[0], ref["reg"].numpy()[0], (S, S), threshold=thr)
    ref_boxes = np.asarray(ref_boxes, np.float32).reshape(-1, 5)
    bm_ref32 = ew.box_match([p32[0]], [ref_boxes], 0.5, device=dev_index)
    bm_ref16 = ew.box_match([p16[0]], [ref_boxes], 0.5, device=dev_index)
    box_match = {
        "benchmarked_vs_exact_fp32_engine": {"images": B, "recall": round(bm["recall"], 5), "precision": round(bm["precision"], 5),
                                             "boxes_benchmarked": int(sum(len(b) for b in p16)), "boxes_exact": int(sum(len(b) for b in p32))},
        "benchmarked_vs_cpu_oracle_image0": {"recall": round(bm_ref16["recall"], 5), "precision": round(bm_ref16["precision"], 5), "boxes_oracle": int(len(ref_boxes))},
        "exact_fp32_engine_vs_cpu_oracle_image0": {"recall": round(bm_ref32["recall"], 5), "precision": round(bm_ref32["precision"], 5),
                                                   "max_abs_box_diff_px": (round(float(np.abs(p32[0][:, :4] - ref_boxes[:, :4]).max()), 6)
                                                                           if p32[0].shape == ref_boxes.shape and len(ref_boxes) else None)},
        "iou_threshold": 0.5, "score_threshold": round(thr, 6), "nms_threshold": 0.3, "wh_bias_shift": 6.0,
        "note": "eval_widerface.evaluate's measure (its 'recall' = detections matching a reference box / reference boxes, its 'precision' "
                "= reference boxes matched by a detection / detections), D2 decode + NMS as get_detections; computed by cf_op_box_match; "
                "synthetic weights with the wh head's bias shifted +6 (the reference's decoders use the size head linearly: a zero-mean head "
                "gives boxes of negative extent)",
    }
    overlap, same_rank, dbox, dscore = [], 0, 0.0, 0.0
    for b in range(B):
        pos32 = {int(c): r for r, c in enumerate(i32[b])}
        both = [(r, pos32[int(c)]) for r, c in enumerate(i16[b]) if int(c) in pos32]
        overlap.append(len(both))
        same_rank += int((i16[b] == i32[b]).sum())
        if both:
            r16, r32 = np.array(both).T
            dbox = max(dbox, float(np.abs(d16[b, r16, :4] - d32[b, r32, :4]).max()))
            dscore = max(dscore, float(np.abs(d16[b, r16, 4] - d32[b, r32, 4]).max()))
    emu = E.forward(cfa.weights.synthetic_state_dict(0), img_u8=host_imgs[:1])
    sg = O.sigmoid_clamp(emu["hm"]).numpy()
    ed, _, ei = O.ctdet_decode(sg, emu["wh"].numpy(), emu["reg"].numpy(), K)
    rms = float(np.sqrt((emu["hm"].numpy() ** 2).mean()))
    aux = {"p32": p32, "ref_boxes": ref_boxes, "thr": thr, "sd_m": sd_m}
    return aux, {
        "box_match": box_match,
        "vs_fp32_parity_engine": {
            "images": B, "topk": K, "index_overlap_mean": round(float(np.mean(overlap)), 2), "index_overlap_min": int(min(overlap)),
            "same_index_same_rank_frac": round(same_rank / (B * K), 4),
            "max_abs_box_diff_map_px": round(dbox, 4), "max_abs_score_diff": round(dscore, 5)},
        "vs_bf16_emulating_oracle": {
            "images": 1, "index_overlap": len(set(i16[0].tolist()) & set(ei[0].tolist())),
            "same_index_same_rank": int((i16[0] == ei[0]).sum()),
            "hm_logit_mean_abs_diff_over_rms": round(float(np.abs(h16["hm"][0] - emu["hm"].numpy()[0]).mean()) / rms, 5),
            "max_abs_score_diff": round(float(np.abs(h16["hm_sigmoid"][0] - sg[0]).max()), 5),
            "note": "image 0 of the timed batch only (the CPU emulation of all 64 takes minutes; tests/test_bf16_parity.py::"
                    "test_batch64_end_to_end_vs_emulation emulates 4 of the 64); kernel-level agreement (99.75-99.99 % of outputs "
                    "bit-identical, layer by layer) is asserted in tests/test_bf16_parity.py; end to end two bf16 pipelines drift apart "
                    "through 1-ulp rounding flips"},
    }


def kernel_profile(eng, cfa, d_in_ptr, B, K, reps):
    """HIP events on the ctx stream around every launch of the forward (cf_profile_forward), `reps` times: per kernel symbol the
    summed ms / algorithmic bytes / flops / launches, and the dominant symbol."""
    agg = {}
    for _ in range(reps):
        for r in eng.profile_forward(d_in_ptr, on_device=True, B=B, in_format=cfa._lib.CF_IN_U8_HWC_BGR, K=K):
            a = agg.setdefault(r["kernel"], dict(kind=r["kind"], ms=0.0, bytes=0.0, flops=0.0, launches=0, layers=set()))
            a["ms"] += r["ms"]; a["bytes"] += r["algo_bytes"]; a["flops"] += r["flops"]; a["launches"] += 1
            a["layers"].add(r["name"])
    tot_ms = sum(a["ms"] for a in agg.values()) / reps
    dom_name, dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
    return agg, tot_ms, dom_name, dom


def tolerance_block(cfa, host_imgs, d_in_ptr, B, S, K, dev_index, aux, profile_reps):
    """The timed batch through the tolerance mode (fp32_split): every head value and the decode against the exact-fp32 engine,
    against the CPU oracle (= the reference's arithmetic) on image 0, BASELINE.json's "box match vs ref" for this mode, and the
    roofline of its dominant kernel."""
    import glob
    import torch
    from oracle import centerface_oracle as O
    from centerface_amd import eval_widerface as ew
    fmt = cfa._lib.CF_IN_U8_HWC_BGR
    res = {}
    roof = None
    for dt in ("fp32", "fp32_split"):
        e = cfa.Engine(S, S, max_batch=B, dtype=dt, device=dev_index)
        e.forward_enqueue(d_in_ptr, on_device=True, B=B, in_format=fmt)
        res[dt] = (e.heads(sigmoid_hm=True), e.decode_topk(K))
        if dt == "fp32_split":
            agg, tot_ms, dom_name, dom = kernel_profile(e, cfa, d_in_ptr, B, K, profile_reps)
            avg_ms, avg_bytes = dom["ms"] / dom["launches"], dom["bytes"] / dom["launches"]
            traffic = None
            try:
                tj = sorted(glob.glob(os.path.join(REPO, "profiles", "*_split_traffic.json")))[-1]
                traffic = json.load(open(tj)).get(dom_name, {}).get("hbm_bytes_per_launch")
            except Exception:                               # noqa: BLE001
                traffic = None
            ach = avg_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "algo_bytes_per_launch": avg_bytes, "layers": sorted(dom["layers"]),
                    "forward_ms_sum_of_kernels": round(tot_ms, 3),
                    "forward_algo_GBps": round(sum(a["bytes"] for a in agg.values()) / profile_reps / (tot_ms * 1e-3) / 1e9, 1),
                    "note": "dominant kernel symbol of the fp32_split forward: algorithmic bytes / HIP-event launch time (one context); traffic = PMC "
                            "FETCH_SIZE x 2 + WRITE_SIZE of the newest committed profiles/*_split_traffic.json; what binds these kernels is LDS + VALU issue "
                            "(DESIGN.md section 4), the HBM fraction is the contract's figure"}
        e.close()
    (h0, (d0, l0, i0)), (h1, (d1, l1, i1)) = res["fp32"], res["fp32_split"]
    out = {"images": B, "topk": K}
    for k in ("hm", "wh", "lm", "reg", "hm_sigmoid"):
        out["max_abs_diff_" + k] = float("%.3g" % np.abs(h0[k] - h1[k]).max())
    out["allclose_1e-3"] = bool(all(np.allclose(h1[k], h0[k], rtol=1e-3, atol=1e-3) for k in ("hm", "wh", "lm", "reg", "hm_sigmoid")))
    same = i0 == i1
    out["same_index_same_rank_frac"] = round(float(same.mean()), 5)
    out["max_abs_box_diff_map_px"] = float("%.3g" % np.abs(d0[..., :4][same] - d1[..., :4][same]).max()) if same.any() else None
    out["max_abs_score_diff"] = float("%.3g" % np.abs(d0[..., 4][same] - d1[..., 4][same]).max()) if same.any() else None
    # ranks that differ: score ties within the head difference (the two modes' rounding decides the order)
    if (~same).any():
        out["score_gap_at_differing_ranks_max"] = float("%.3g" % np.abs(d0[..., 4][~same] - d1[..., 4][~same]).max())
    # ---- against the CPU oracle (the reference's fp32 arithmetic), image 0 of the timed batch: north_star's tolerance itself
    ref = O.forward(O.to_torch_sd(cfa.weights.synthetic_state_dict(0)), torch.from_numpy(O.preprocess(host_imgs[0])))
    ref_hm = O.sigmoid_clamp(ref["hm"]).numpy()
    vs_oracle = {"image": 0}
    for k in ("hm", "wh", "lm", "reg"):
        vs_oracle["max_abs_diff_" + k] = float("%.3g" % np.abs(h1[k][:1] - ref[k].numpy()).max())
    vs_oracle["max_abs_diff_hm_sigmoid"] = float("%.3g" % np.abs(h1["hm_sigmoid"][:1] - ref_hm).max())
    vs_oracle["allclose_1e-3"] = bool(all(np.allclose(h1[k][:1], ref[k].numpy(), rtol=1e-3, atol=1e-3) for k in ("hm", "wh", "lm", "reg")) and
                                      np.allclose(h1["hm_sigmoid"][:1], ref_hm, atol=1e-3, rtol=0))
    rdet, _, rinds = O.ctdet_decode(ref_hm, ref["wh"].numpy(), ref["reg"].numpy(), K, ref["lm"].numpy())
    same0 = i1[0] == rinds[0]
    vs_oracle["same_index_same_rank"] = int(same0.sum())
    vs_oracle["index_overlap"] = len(set(i1[0].tolist()) & set(rinds[0].tolist()))
    if same0.any():
        vs_oracle["max_abs_box_diff_map_px"] = float("%.3g" % np.abs(d1[0][same0][:, :4] - rdet[0][same0][:, :4]).max())
        vs_oracle["max_abs_score_diff"] = float("%.3g" % np.abs(d1[0][same0][:, 4] - rdet[0][same0][:, 4]).max())
    # ---- "box match vs ref" for this mode: the same measure, threshold and shifted weights as parity.box_match
    box_match = None
    if aux is not None:
        e = cfa.Engine(S, S, max_batch=B, dtype="fp32_split", device=dev_index, weights=aux["sd_m"])
        e.forward_enqueue(d_in_ptr, on_device=True, B=B, in_format=fmt)
        psp = [b for b, _ in e.decode_threshold(aux["thr"], 0.3, mode="d2")]
        e.close()
        bm = ew.box_match(psp, aux["p32"], 0.5, device=dev_index)
        bmo = ew.box_match([psp[0]], [aux["ref_boxes"]], 0.5, device=dev_index)
        rb = aux["ref_boxes"]
        box_match = {"tolerance_vs_exact_fp32_engine": {"images": B, "recall": round(bm["recall"], 5), "precision": round(bm["precision"], 5),
                                                        "boxes_tolerance": int(sum(len(b) for b in psp)), "boxes_exact": int(sum(len(b) for b in aux["p32"]))},
                     "tolerance_vs_cpu_oracle_image0": {"recall": round(bmo["recall"], 5), "precision": round(bmo["precision"], 5), "boxes_oracle": int(len(rb)),
                                                        "max_abs_box_diff_px": (round(float(np.abs(psp[0][:, :4] - rb[:, :4]).max()), 6)
                                                                                if psp[0].shape == rb.shape and len(rb) else None)},
                     "iou_threshold": 0.5, "score_threshold": round(aux["thr"], 6), "nms_threshold": 0.3, "wh_bias_shift": 6.0}
    return out, vs_oracle, box_match, roof


def main():
    args = parse()
    if args.cpu_baseline_child:
        _cpu_baseline_child(args.cpu_baseline_child)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    import centerface_amd as cfa

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback path exists)", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    B, S, K = args.batch, args.size, args.topk
    # synthetic input: uint8 [B,S,S,3] BGR, numpy default_rng(rank), resident in HBM
    rng = np.random.default_rng(rank)
    host_imgs = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    d_in = torch.from_numpy(host_imgs).to(dev)
    # the timed steps rotate through --input-buffers distinct batches (buffer 0 = host_imgs, the one the parity block uses)
    d_ins = [d_in] + [torch.from_numpy(np.random.default_rng(1000 * (j + 1) + rank).integers(0, 256, (B, S, S, 3), dtype=np.uint8)).to(dev)
                      for j in range(max(1, args.input_buffers) - 1)]
    d_in_ptrs = [t.data_ptr() for t in d_ins]
    D = args.depth if args.depth > 0 else (3 if B * S * S <= (8 << 20) else 2)
    outs = [{"dets": torch.empty((B, K, 6), dtype=torch.float32, device=dev),
             "lms": torch.empty((B, K, 10), dtype=torch.float32, device=dev),
             "inds": torch.empty((B, K), dtype=torch.int64, device=dev),
             "all": torch.empty((world * B, K, 16), dtype=torch.float32, device=dev)} for _ in range(D)]
    out = outs[0]
    torch.cuda.synchronize()

    ring = cfa.EngineRing(S, S, depth=D, max_batch=B, dtype=args.dtype, device=local_rank)   # contexts of alternating stream priority
    engs = ring.engines
    eng = engs[0]
    gather, comms = "none", []
    hard_exit = False            # a helper thread is stuck inside ncclCommInitRank: leave through os._exit after the JSON line

    def close_comms():
        for cm in comms:
            cm.close()
        del comms[:]
    fallback = None
    if world > 1 or args.exercise_gather_path:
        gather = "torch" if args.gather == "torch" else "cf"
        if gather == "cf":
            # ONE RCCL communicator per rank (rendezvous of the 128-byte id over the torch.distributed store), shared by the
            # rank's contexts.  Creation and the first gather are checked against a deadline, and the verdict is agreed over
            # the TCP store -- never over NCCL, which is the thing under suspicion.
            ok = 1
            try:
                uid = cfa.distributed.broadcast_unique_id() if world > 1 else cfa.distributed.unique_id()
                # ncclCommInitRank blocks until every rank has joined: run it on a helper thread against the same deadline, so a
                # rank that never arrives costs a fallback, not the run (the stuck thread is abandoned; see hard_exit below)
                import threading
                box = {}

                def create():
                    try:
                        box["comm"] = cfa.distributed.Comm(engs[0], rank, world, uid)
                    except Exception as exc2:                          # noqa: BLE001
                        box["err"] = exc2
                th = threading.Thread(target=create, daemon=True)
                th.start()
                th.join(max(5.0, abs(args.gather_timeout)))
                if th.is_alive():
                    ok, hard_exit = 0, True
                    print("rank %d: cf_comm_create did not return within %.0f s" % (rank, max(5.0, abs(args.gather_timeout))), file=sys.stderr)
                elif "err" in box:
                    raise box["err"]
                else:
                    comms.append(box["comm"])
            except Exception as exc:                                   # noqa: BLE001
                ok = 0
                print("rank %d: cf_comm_create failed (%s)" % (rank, exc), file=sys.stderr)
            if not cfa.distributed.agree(ok, rank, world, "cf_comm_created"):
                fallback = "cf_comm_create failed on some rank"
            else:
                try:
                    if args.debug_stall_gather_ms > 0:
                        comms[0].debug(0, args.debug_stall_gather_ms)
                    # The FIRST RCCL collective of the run is the shard agreement (cf_comm_set_shard: a 2-int all-gather, enqueued,
                    # never waited for inside the library); it is polled here against the deadline.  Only then are record gathers
                    # enqueued -- one collective per step, each slot carrying (B, K, step) in its header -- and polled the same way.
                    comms[0].set_shard(B, K)
                    ok = 1 if (args.gather_timeout >= 0 and comms[0].wait(args.gather_timeout)) else 0      # < 0: forced fallback (tests)
                    if ok:
                        for e, o in zip(engs, outs):               # warm-up gather of every context, in step order
                            e.forward_enqueue(d_in.data_ptr(), on_device=True, B=B, in_format=cfa._lib.CF_IN_U8_HWC_BGR)
                            comms[0].gather_topk_device(K, o["all"].data_ptr(), engine=e)
                        ok = 1 if comms[0].wait(args.gather_timeout) else 0
                except Exception as exc:                                   # noqa: BLE001
                    ok = 0
                    print("rank %d: first cf_gather_topk failed (%s)" % (rank, exc), file=sys.stderr)
                if not cfa.distributed.agree(ok, rank, world, "cf_first_gather"):
                    fallback = "first RCCL gather did not complete within %.0f s on some rank" % args.gather_timeout
            if fallback:
                if args.gather == "cf":
                    print("rank %d: %s" % (rank, fallback), file=sys.stderr)
                    sys.exit(4)
                for cm in comms:
                    cm.abort()
                del comms[:]
                gather = "torch"
                for e in engs[1:]:                                         # the fallback is the conservative schedule: one context
                    e.close()
                engs, outs, D = engs[:1], outs[:1], 1
    step = make_step(cfa, engs, d_in_ptrs, B, K, outs, gather, comms if gather == "cf" else None)
    flush_c_stdio()          # librccl prints a version banner through C stdio when a communicator is created: get it out now

    def fence():
        for e in engs:
            e.synchronize()
        for cm in comms:
            cm.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    wins = time_windows(step, fence, args.steps, max(1, args.repeats), reduce_max)
    med = float(np.median(wins))

    result = None
    parity_aux = None
    if rank == 0:
        value = world * B * args.steps / med
        # ---- per-kernel profile (HIP events on the ctx stream around every launch)
        agg, tot_ms, dom_name, dom = kernel_profile(eng, cfa, d_in.data_ptr(), B, K, args.profile_reps)
        avg_ms = dom["ms"] / dom["launches"]
        avg_bytes = dom["bytes"] / dom["launches"]
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        try:
            import glob
            tj = args.traffic_json or sorted(glob.glob(os.path.join(REPO, "profiles", "traffic_*.json")))[-1]
            with open(tj) as f:
                traffic = json.load(f).get(dom_name, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        kinds = {}
        for a in agg.values():
            k = kinds.setdefault(a["kind"], dict(ms=0.0, bytes=0.0))
            k["ms"] += a["ms"] / args.profile_reps; k["bytes"] += a["bytes"] / args.profile_reps
        # ---- the roof that actually binds the fused kernels: VALU instruction issue.  Dynamic instruction counts per launch
        # (rocprofv3 --pmc SQ_INSTS_*, committed as profiles/*_valu_counts.json by tools/valu_bound.py for THIS workload) x the
        # committed issue costs, against this run's launch durations (HIP events) and the committed rocprofv3 durations.
        valu_issue = None
        try:
            import glob
            vj = sorted(glob.glob(os.path.join(REPO, "profiles", "*_valu_counts.json")))[-1]
            vc = json.load(open(vj))
            if (B, S, args.dtype) == (64, 640, "bf16"):
                per, sb, sm, sr, sh = [], 0.0, 0.0, 0.0, 0.0
                have_hw = True
                for name, a in agg.items():
                    kc = vc["kernels"].get(name)
                    if kc is None:
                        continue
                    n = a["launches"] // args.profile_reps
                    ms = a["ms"] / args.profile_reps
                    sb += kc["bound_us"] * n * 1e-3; sm += ms; sr += kc["rocprof_avg_us"] * n * 1e-3
                    hw = kc.get("bound_hw_us")
                    have_hw = have_hw and hw is not None
                    sh += (hw or 0.0) * n * 1e-3
                    per.append({"kernel": name, "launches": n, "bound_ms": round(kc["bound_us"] * n * 1e-3, 4), "measured_ms": round(ms, 4),
                                "rocprof_ms": round(kc["rocprof_avg_us"] * n * 1e-3, 4), "frac": round(kc["bound_us"] * n * 1e-3 / ms, 3) if ms > 0 else None,
                                "bound_hw_ms": round(hw * n * 1e-3, 4) if hw is not None else None,
                                "frac_hw": round(hw * n * 1e-3 / ms, 3) if (hw is not None and ms > 0) else None})
                per.sort(key=lambda r: -r["measured_ms"])
                valu_issue = {"bound_ms": round(sb, 4), "measured_ms": round(sm, 4), "rocprof_ms": round(sr, 4),
                              "frac": round(sb / sm, 4) if sm > 0 else None, "frac_vs_rocprof": round(sb / sr, 4) if sr > 0 else None,
                              "bound_hw_ms": round(sh, 4) if have_hw else None, "frac_hw": round(sh / sm, 4) if (have_hw and sm > 0) else None,
                              "frac_hw_vs_rocprof": round(sh / sr, 4) if (have_hw and sr > 0) else None,
                              "cost_table": vc.get("cost_cycles_per_wave_instruction"), "cost_table_hw": vc.get("cost_hw_cycles_per_wave_instruction"),
                              "counts": os.path.basename(vj), "counts_are": "committed dynamic instruction counts of this workload (a constant of the repo, not of this run); the measured_ms they are divided by are this run's",
                              "per_kernel": per[:12],
                              "note": "TWO cost tables.  bound_ms: sum over the forward's kernels of (wave-instruction counts x MEASURED issue cost, wall time of a dense "
                                      "stream of that class expressed in cycles of a nominal 2.4 GHz: transcendental 9.45, packed 5.65, dot2c 4.53, VOP3 4.5, plain 3.1; "
                                      "LDS / VMEM / MFMA issue slots) / 1024 SIMDs / 2.4 GHz -- what this instruction mix costs in wall time.  bound_hw_ms: VALU "
                                      "instructions only at ARCHITECTURAL rates (transcendental 8, packed / dot2 4, everything else 2 shader cycles at 2.4 GHz): the "
                                      "hardware guide's floor, which no sustained stream reaches (tools/ub_clock_probe.hip: dense v_fma_f32 = 2.28 shader cycles at a "
                                      "power-capped 1.94 GHz).  measured = HIP events of this run (one context), rocprof = committed rocprofv3 --kernel-trace averages"}
        except Exception:                                   # noqa: BLE001
            valu_issue = None
        dom_valu = None
        if valu_issue:
            dom_valu = next((r["frac"] for r in valu_issue["per_kernel"] if r["kernel"] == dom_name), None)
        hbm_frac = achieved / HBM_PEAK_GBS
        roofline = {
            "bound": "valu_issue" if (dom_valu is not None and dom_valu > hbm_frac) else "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "avg_launch_ms": round(avg_ms, 4), "algo_bytes_per_launch": avg_bytes,
            "launches_per_forward": dom["launches"] // args.profile_reps, "layers": sorted(dom["layers"]),
            "valu_issue_frac": dom_valu, "valu_issue": valu_issue,
            "forward_ms_sum_of_kernels": round(tot_ms, 3),
            # the fused kernels are VALU-issue bound (Swish: two transcendentals per expanded element), not HBM- or
            # MFMA-bound: DESIGN.md section 4, profiles/r01_valu_microbench.md; whole-forward algorithmic rate:
            "forward_algo_GBps": round(sum(a["bytes"] for a in agg.values()) / args.profile_reps / (tot_ms * 1e-3) / 1e9, 1),
            "forward_TFLOPs": round(sum(a["flops"] for a in agg.values()) / args.profile_reps / (tot_ms * 1e-3) / 1e12, 1),
            "by_kind": {k: {"ms": round(v["ms"], 3), "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
                        for k, v in kinds.items()},
        }
        result = {
            "metric": "images/sec at 640x640 batch inference", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * med / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s: batch=%d %dx%d %s on 1 MI355X per rank, forward + top-%d peak decode%s; "
                                   "synthetic uint8 images resident in HBM, calibrated synthetic weights (seed 0); "
                                   "%d context(s) per GPU, batch k on context k mod %d"
                                   % ("BASELINE configs[1]" if (B, S, K, args.dtype) == (64, 640, 100, "bf16") else
                                      "BASELINE configs[4] per-GPU shard" if (S, K) == (1280, 1000) else "custom (not a BASELINE config)",
                                      B, S, S, args.dtype, K, " + RCCL all-gather of boxes" if world > 1 else "", D, D),
                       "batch_per_gpu": B, "global_batch": B * world, "image": [S, S], "topk": K,
                       "parallelism": "dp%d" % world,
                       "contexts_per_gpu": len(engs), "input_buffers": len(d_ins),
                       "gather": {"none": None, "cf": "cf_gather_topk (C ABI: decode on the context's %s stream, ONE ncclAllGather per step on the rank's one gather stream / one communicator; slot header validated on the device)" % ("decode" if len(engs) < 3 else "main"),
                                  "torch": "torch.distributed.all_gather_into_tensor"}[gather],
                       "gather_fallback": fallback},
            "windows": {"n": len(wins), "steps_each": args.steps, "median_ms": round(1e3 * med, 5),
                        "min_ms": round(1e3 * min(wins), 5), "max_ms": round(1e3 * max(wins), 5),
                        "value_min": round(world * B * args.steps / max(wins), 1), "value_max": round(world * B * args.steps / min(wins), 1)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_extras:
            # ---- PCIe-inclusive: the batch starts in pinned host memory every step (H2D on the copy stream, overlapped)
            pinned = torch.from_numpy(host_imgs).pin_memory()
            hview = pinned.numpy()

            def step_h2d():                     # one context: this step is PCIe-bound (78.6 MB per batch), a second batch in flight buys nothing
                eng.forward_enqueue(hview)
                eng.decode_topk_device(K, out["dets"].data_ptr(), out["lms"].data_ptr(), out["inds"].data_ptr())
            for _ in range(3):
                step_h2d()
            wh = time_windows(step_h2d, fence, args.steps, 7)
            result["value_with_h2d"] = {"value": round(B * args.steps / float(np.median(wh)), 1), "unit": "images/s",
                                        "note": "pinned host batch (%.1f MB) copied every step on the copy stream, overlapped with the "
                                                "previous forward, one context (PCIe-bound: ~46 GB/s); median of 7 windows" % (host_imgs.nbytes / 1e6)}
            if D > 1:
                # ---- one context only: every step on one stream chain (what a caller without the ring gets)
                st1 = make_step(cfa, eng, d_in_ptrs, B, K, out)
                for _ in range(3):
                    st1()
                w1 = time_windows(st1, fence, args.steps, 7)
                result["value_one_context"] = {"value": round(B * args.steps / float(np.median(w1)), 1), "unit": "images/s",
                                               "note": "the same steps on a single context (one batch in flight); median of 7 windows"}
            parity_aux, result["parity"] = parity_block(cfa, eng, host_imgs, d_in.data_ptr(), B, S, K, local_rank)
    eng_closed = False
    if rank == 0 and world == 1 and not args.no_extras and args.dtype == "bf16":
        close_comms()
        for e in engs:
            e.close()
        eng_closed = True

        def mode_rate(dtype):
            """The headline's schedule (two contexts round-robin, rotating inputs) and the one-context schedule in another mode."""
            r2 = cfa.EngineRing(S, S, depth=D, max_batch=B, dtype=dtype, device=local_rank)
            o2 = outs if len(outs) == len(r2.engines) else [outs[0]] * len(r2.engines)
            st = make_step(cfa, r2.engines, d_in_ptrs, B, K, o2)

            def fence2():
                for e in r2.engines:
                    e.synchronize()
                torch.cuda.synchronize()
            for _ in range(3):
                st()
            w = time_windows(st, fence2, 6, 7)
            stb = make_step(cfa, r2.engines[0], d_in_ptrs, B, K, out)
            for _ in range(2):
                stb()
            wb = time_windows(stb, fence2, 5, 5)
            r2.close()
            return round(B * 6 / float(np.median(w)), 1), round(B * 5 / float(np.median(wb)), 1), len(r2.engines)
        # ---- the other single-GPU BASELINE configs, measured in this run (parity-test cases in the contract's sense: not the headline)
        try:
            result["other_configs"] = other_configs_block(cfa, local_rank)
        except Exception as exc:                                # noqa: BLE001  (never costs the headline its line)
            result["other_configs"] = {"error": repr(exc)[:300]}
        try:
            result["standalone_depthwise"] = standalone_depthwise_block(cfa, d_in.data_ptr(), B, S, K, local_rank)
        except Exception as exc:                                # noqa: BLE001
            result["standalone_depthwise"] = {"error": repr(exc)[:300]}
        # ---- tolerance mode: the mode that meets north_star's "box/score within 1e-3 of the reference" at speed
        v, v1, nctx = mode_rate("fp32_split")
        tol_exact, tol_oracle, tol_bm, tol_roof = tolerance_block(cfa, host_imgs, d_in.data_ptr(), B, S, K, local_rank, parity_aux, max(1, args.profile_reps))
        result["tolerance_mode"] = {
            "value": v, "unit": "images/s", "batch": B, "value_one_context": v1, "dtype": "fp32_split",
            "arithmetic": "fp32 storage in HBM and LDS; every GEMM product (stem, expand, project, neck, heads) as a split-bf16 product on the bf16 "
                          "matrix pipe: x = hi + lo (bf16 pairs, 16 mantissa bits), w.x = w_hi.x_hi + w_lo.x_hi + w_hi.x_lo on v_mfma_f32_32x32x8_bf16_1k, fp32 "
                          "accumulate; Swish, depthwise taps, residual adds and epilogues in fp32",
            "vs_exact_fp32_engine": tol_exact, "vs_cpu_oracle_image0": tol_oracle, "box_match": tol_bm, "roofline": tol_roof,
            "note": "%d context(s) round-robin like the headline; median of 7 windows of 6 steps; the reference-golden and oracle tests at "
                    "rtol = atol = 1e-3 run on this mode too (tests/test_gpu_parity.py, EXACT)" % nctx}
        v, v1, nctx = mode_rate("fp32")
        result["exact_fp32_mode"] = {"value": v, "unit": "images/s", "batch": B, "value_one_context": v1,
                                     "note": "fp32 storage + exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bit-equal to an fmaf chain), the bit-level "
                                             "test mode; %d context(s) round-robin; median of 7 windows of 6 steps" % nctx}
    if not eng_closed:
        close_comms()
        for e in engs:
            e.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_cpu_baseline:                # rank 0, every N: the other ranks have left by now and the host cores are idle
            result["cpu_baseline"] = cpu_baseline(args.cpu_seconds, S, K, host_imgs)
        else:
            result["cpu_baseline"] = None
        flush_c_stdio()
        print(json.dumps(result), flush=True)       # the one JSON line, last on stdout
    if hard_exit:                                   # a thread is still blocked inside ncclCommInitRank: skip interpreter teardown
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()

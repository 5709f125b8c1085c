#!/usr/bin/env python3
"""A/B of the two-lane schedule (cf_forward_lanes) against the free-running pair of contexts (EngineRing), B = 64, 640x640,
forward + top-100 decode, inputs resident in HBM; also checks that both schedules return identical boxes.
`python tools/lanes_probe.py [--steps 40] [--windows 9]`; cut points via CF_LANE_CUT1 / CF_LANE_CUT2 (op-name prefixes).
Needs the EXPERIMENTS build of the library (cf_forward_lanes is not in the product ABI since round 4):
`make -C lightweight-face-detection-centernet_amd/csrc EXP=1` and `CF_LIB=.../libcenterface_hip_exp.so python tools/lanes_probe.py`."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40); ap.add_argument("--windows", type=int, default=9)
    ap.add_argument("--batch", type=int, default=64); ap.add_argument("--size", type=int, default=640); ap.add_argument("--topk", type=int, default=100)
    a = ap.parse_args()
    import torch
    B, S, K = a.batch, a.size, a.topk
    rng = np.random.default_rng(0)
    d_in = [torch.from_numpy(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).cuda() for _ in range(2)]
    ring = cfa.EngineRing(S, S, depth=2, max_batch=B, dtype="bf16")
    engs = ring.engines
    outs = [{"dets": torch.empty((B, K, 6), dtype=torch.float32, device="cuda"), "lms": torch.empty((B, K, 10), dtype=torch.float32, device="cuda"),
             "inds": torch.empty((B, K), dtype=torch.int64, device="cuda")} for _ in range(2)]

    def dec(i):
        engs[i].decode_topk_device(K, outs[i]["dets"].data_ptr(), outs[i]["lms"].data_ptr(), outs[i]["inds"].data_ptr())

    def fence():
        for e in engs:
            e.synchronize()
        torch.cuda.synchronize()

    def run_ring(n, k0=0):
        for k in range(k0, k0 + n):
            i = k % 2
            engs[i].forward_enqueue(d_in[i].data_ptr(), on_device=True, B=B, in_format=0)
            dec(i)

    state = {"prev": None}

    def run_lanes(n, k0=0):
        for k in range(k0, k0 + n):
            i = k % 2
            p = state["prev"]
            engs[i].forward_lanes_enqueue(engs[p] if p is not None else None, d_in[i].data_ptr(), B)
            if p is not None:
                dec(p)
            state["prev"] = i

    def flush_lanes():
        p = state["prev"]
        if p is not None:
            engs[p].forward_lanes_flush(); dec(p); state["prev"] = None

    # results: ring vs lanes
    run_ring(2); fence()
    want = [{k: v.clone() for k, v in o.items()} for o in outs]
    for o in outs:
        for v in o.values():
            v.zero_()
    run_lanes(2); flush_lanes(); fence()
    same = all(torch.equal(outs[i][k], want[i][k]) for i in range(2) for k in ("dets", "lms", "inds"))
    print("identical results:", same)

    def bench(run, flush=None):
        run(6); (flush or (lambda: None))(); fence()
        ts = []
        for _ in range(a.windows):
            fence(); t0 = time.perf_counter(); run(a.steps); (flush or (lambda: None))(); fence(); ts.append(time.perf_counter() - t0)
        return B * a.steps / float(np.median(ts)), 1e3 * float(np.median(ts)) / a.steps
    print("ring  : %.0f img/s  %.4f ms/step" % bench(run_ring))
    print("lanes : %.0f img/s  %.4f ms/step  (cuts %s / %s)" % (bench(run_lanes, flush_lanes) + (os.environ.get("CF_LANE_CUT1", "layer2.0"), os.environ.get("CF_LANE_CUT2", "layer4.0"))))
    print("ring  : %.0f img/s  %.4f ms/step" % bench(run_ring))


if __name__ == "__main__":
    main()

"""The C ABI used from plain C (examples/detect.c): compiles and links as C99 on the CPU box; on a GPU box it
runs end to end (weights file -> cf_create / cf_load_weights / cf_forward / cf_decode_topk) and must report the
same best cell per image as the Python host path on the same bytes."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "lightweight-face-detection-centernet_amd")


def _build(tmp_path):
    import centerface_amd as cfa
    cfa._lib.build()
    exe = str(tmp_path / "detect")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(REPO, "include"),
           os.path.join(REPO, "examples", "detect.c"), "-o", exe, "-L" + PKG, "-lcenterface_hip", "-Wl,-rpath," + PKG]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_header_is_c99_and_example_links(tmp_path):
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                    os.path.join(REPO, "include", "centerface_hip.h")], check=True, capture_output=True)
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_example_matches_python_host(tmp_path):
    import centerface_amd as cfa
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import export_weights
    exe = _build(tmp_path)
    wfile = str(tmp_path / "w.bin")
    export_weights.export(wfile, cfa.weights.synthetic_state_dict(0))
    H, W, B = 64, 96, 2
    out = subprocess.run([exe, wfile, str(H), str(W), str(B), "gather"], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert out[-2].startswith("gathered %d x 10 records over RCCL" % B), out[-2]     # cf_comm_* / cf_gather_topk from plain C
    assert out[-1].startswith("grouped communicator: gathered %d x 10 records" % B), out[-1]      # cf_comm_create_all / cf_comm_abort
    dry = subprocess.run([exe, wfile, str(H), str(W), "4", "dryrun", "4"], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert dry[-1].startswith("dry run: 4 ranks x 1 image gathered in rank-major order"), dry[-1]      # cf_comm_create_loopback from plain C: no RCCL
    out = [l for l in out if l.startswith("image ")]                                    # (librccl prints its version banner on stdout)
    assert len(out) == B
    # the same bytes as detect.c's LCG
    s, vals = 12345, np.empty(B * H * W * 3, np.uint8)
    for i in range(vals.size):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[i] = s >> 24
    eng = cfa.Engine(H, W, max_batch=B, dtype="bf16")
    eng.forward_enqueue(vals.reshape(B, H, W, 3))
    dets, _, inds = eng.decode_topk(10)
    for b, line in enumerate(out):
        cell = int(line.split("at cell")[1].split(",")[0])
        score = float(line.split("best score")[1].split("at")[0])
        assert cell == int(inds[b, 0]) and abs(score - float(dets[b, 0, 4])) < 1e-4, (line, inds[b, 0], dets[b, 0, 4])
    eng.close()

"""The run-time switches of the RELEASE library (csrc/cf_common.h::cf_env_int; `strings libcenterface_hip.so | grep ^CF_` lists
exactly these) -- each exercised here, in a child process (they are read once per process):

* ``CF_DW_MATRIX=0``     matrix-core depthwise off: stem, layer1.1, 2.0, 2.1, 4.0, 4.1, 5.1, 6.0 run the v_dot2c kernel family
                         instead.  A different summation order inside a depthwise row, same storage points: the per-kernel and
                         layer-by-layer bf16 parity tests must pass on it.
* ``CF_XCD_ORDER=0``     XCD-aware tile order of the fused stem off: a pure re-mapping of workgroups, results bit-identical.
* ``CF_DECODE_OVERLAP=0`` device-output decode on the main stream instead of the decode stream: results bit-identical.
* ``CF_F4_VARIANT=1``    every fp32 block on the second-generation kernel: tests/test_gpu_parity.py::
                         test_fp32_second_generation_kernel_on_every_block_shape.

Every other A/B switch of the kernel-variant tables exists only in an experiments build (``make -C csrc EXP=1``,
-DCF_EXPERIMENTS -> libcenterface_hip_exp.so); the release library does not contain the variants.
"""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import centerface_amd as cfa

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DIGEST = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import centerface_amd as cfa
h = hashlib.sha256()
kernels = set()
for (S, B, K) in ((320, 8, 100), (160, 3, 40)):
    imgs = np.random.default_rng(S).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    eng = cfa.Engine(S, S, max_batch=B, dtype="bf16")
    for rep in range(3):                                   # eager, capture, replay
        eng.forward_enqueue(imgs)
        d, l, i = eng.decode_topk(K)
    h.update(d.tobytes()); h.update(l.tobytes()); h.update(i.tobytes())
    dd = eng.device_alloc(B * K * 6 * 4); dl = eng.device_alloc(B * K * 10 * 4); di = eng.device_alloc(B * K * 8)
    eng.forward_enqueue(imgs)
    eng.decode_topk_device(K, dd, dl, di)                  # the CF_DECODE_OVERLAP path
    eng.synchronize()
    d2 = np.empty((B, K, 6), np.float32); eng.memcpy_d2h(d2, dd)
    h.update(d2.tobytes())
    assert np.array_equal(d2, d)
    kernels |= {r["kernel"].split("<")[0] for r in eng.profile_forward(imgs, K=K)}
    eng.close()
print("DIGEST", h.hexdigest())
print("KERNELS", " ".join(sorted(kernels)))
'''

PARITY = r'''
import os, sys
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, "tests"))
import test_bf16_parity as T
for blk in T.BLOCKS:
    T.test_production_mbconv_instances_vs_emulation(blk)
T.test_bf16_engine_layer_by_layer_teacher_forced((96, 128), 3)
T.test_bf16_engine_layer_by_layer_teacher_forced((480, 640), 2)
print("PARITY ok")
'''


def _child(code, **env):
    r = subprocess.run([sys.executable, "-c", code % {"repo": REPO}], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def _field(out, tag):
    return [ln.split(" ", 1)[1] for ln in out.splitlines() if ln.startswith(tag + " ")][0]


def test_release_library_reads_only_the_product_switches():
    """The product library's strings hold exactly the four CF_* names (and CF_LIB / CF_STAGE_THREADS are Python-side)."""
    blob = open(cfa._lib.LIB_PATH, "rb").read()
    import re
    names = sorted({m.decode() for m in re.findall(rb"\x00(CF_[A-Z0-9_]{3,})\x00", blob)})
    assert names == ["CF_DECODE_OVERLAP", "CF_DW_MATRIX", "CF_F4_VARIANT", "CF_XCD_ORDER"], names
    L = cfa._lib.lib()
    assert not hasattr(L, "cf_forward_lanes")               # experiments build only


def test_tile_order_and_decode_stream_switches_are_bit_identical():
    base = _child(DIGEST)
    assert "stem0_mx_kernel" in _field(base, "KERNELS")
    for env in ({"CF_XCD_ORDER": "0"}, {"CF_DECODE_OVERLAP": "0"}, {"CF_XCD_ORDER": "0", "CF_DECODE_OVERLAP": "0"}):
        out = _child(DIGEST, **env)
        assert _field(out, "DIGEST") == _field(base, "DIGEST"), env


def test_matrix_core_depthwise_off_runs_the_dot2c_family_and_keeps_parity():
    out = _child(DIGEST, CF_DW_MATRIX="0")
    ks = _field(out, "KERNELS")
    assert "_mx" not in ks and "stem0_px_kernel" in ks and "mbconv_px_kernel" in ks and "expdw_px_kernel" in ks, ks
    assert "PARITY ok" in _child(PARITY, CF_DW_MATRIX="0")

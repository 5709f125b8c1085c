"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/centerface_hip.h declares, reports errors instead of falling back, and the host
logic (schema, sharding, transform) is right.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

import centerface_amd as cfa

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "centerface_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = cfa._lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 28
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(cfa._lib.EXPORTS) == declared
    assert L.cf_version() == 100
    assert L.cf_strerror(-5).decode().startswith("state_dict")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cfa._lib.CenterFaceError) as e:
        cfa.Engine(64, 64)
    assert e.value.code == -3          # CF_EHIP: fails loudly, no silent CPU path
    with pytest.raises((cfa._lib.CenterFaceError, ValueError)):
        cfa.ops.conv_pw(np.zeros((1, 8, 4, 4), np.float32), np.zeros((8, 8), np.float32))


def test_argument_validation_happens_before_any_gpu_work():
    L = cfa._lib.lib()
    h = ctypes.c_void_p()
    assert L.cf_create(0, 1, 100, 64, 0, 0, ctypes.byref(h)) == -1        # H not a multiple of 32
    assert b"multiples of 32" in L.cf_last_error(None)
    assert L.cf_create(0, 0, 64, 64, 0, 0, ctypes.byref(h)) == -1
    assert L.cf_create(0, 1, 64, 64, 7, 0, ctypes.byref(h)) == -1         # unknown dtype


def test_schema_matches_reference_checkpoint_layout():
    sch = cfa.schema.state_dict_schema()
    assert len(sch) == 94
    assert sum(int(np.prod(s)) for s in sch.values()) == 1308126
    assert sch["layer0.0.conv.0.1.weight"] == (32, 1, 3, 3) and sch["layer0.0.conv.1.weight"] == (16, 32, 1, 1)
    assert sch["layer2.0.conv.1.1.weight"] == (144, 1, 5, 5)
    assert sch["up1.conv.0.weight"] == (24, 96, 1, 1) and sch["lm.1.weight"] == (10, 24, 1, 1)
    sd = cfa.weights.synthetic_state_dict(3)
    assert list(sd) == list(sch)
    bad = dict(sd); bad["extra"] = np.zeros(1, np.float32)
    with pytest.raises(ValueError):
        cfa.weights.validate_state_dict(bad)
    bad = dict(sd); bad["hm.0.bias"] = np.zeros(25, np.float32)
    with pytest.raises(ValueError):
        cfa.weights.validate_state_dict(bad)


def test_shard_range_partitions_in_order():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [cfa.distributed.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        cfa.distributed.shard_range(4, 2, 2)


def test_transform_table(golden):
    g = golden("decode_d1")
    face = object.__new__(cfa.CenterFace)
    for (h, w), ref in zip(g["tf_in"], g["tf_out"]):
        assert np.array_equal(np.asarray(cfa.CenterFace.transform(face, int(h), int(w)), np.float64), ref)


def test_wider_result_format(tmp_path):
    """demo.py:81-87: path line, count line, then 'x y w h score' with w = x2 - x1 + 1."""
    dets = np.array([[10.0, 20.0, 29.0, 59.0, 0.98765], [0.0, 0.0, 5.5, 7.25, 0.05]], np.float32)
    txt = cfa.demo.format_wider_result("0--Parade/0_Parade_marchingband_1_465.jpg", dets)
    assert txt == "0--Parade/0_Parade_marchingband_1_465.jpg\n2\n10.0 20.0 20.0 40.0 0.988\n0.0 0.0 6.5 8.2 0.050\n"
    p = cfa.demo.write_wider_result(str(tmp_path), "0--Parade", "img_1", np.empty((0, 5), np.float32))
    assert open(p).read() == "0--Parade/img_1.jpg\n0\n" and p.endswith("0--Parade/img_1.txt")


def test_fast_floor_division_equals_numpy_floor_divide():
    """CenterFace._floordiv must be `a // scale` (centerface.py:56,58) bit for bit: random coordinates, values at and one
    ulp around exact integer multiples of the scale, negatives (landmarks left of the image), signed zeros, scale 1."""
    import centerface_amd as cfa
    rng = np.random.default_rng(0)
    for trial in range(40):
        h = int(rng.integers(33, 1300))
        s = (int(np.ceil(h / 32) * 32) / h) if trial else 1.0
        s32 = np.float32(s)
        n = rng.integers(0, 1300, 20000).astype(np.float32)
        near = (n * s32).astype(np.float32)
        a = np.concatenate([rng.uniform(-50, 1300, 50000).astype(np.float32), near, np.nextafter(near, np.float32(np.inf)),
                            np.nextafter(near, np.float32(-np.inf)), -near[:500], np.array([0.0, -0.0], np.float32)])
        ref, got = a // s, cfa.CenterFace._floordiv(a, s)
        assert got.dtype == np.float32 and np.array_equal(ref, got) and np.array_equal(np.signbit(ref), np.signbit(got)), s

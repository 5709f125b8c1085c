"""GPU parity tests: every HIP kernel, called through the C ABI, against the reference's golden
vectors (tests/golden, produced by importing the reference) and against the oracle on seeded inputs.

Tolerances (BASELINE.json north_star): top-k indices bit-exact, box/score fp32 within 1e-3.
  * fp32 mode (CF_F32: fp32 storage + exact-fp32 MFMA) is the parity mode: 1e-3 abs/rel end to end,
    ~1e-5 per op.  Decode kernels are integer/selection work on given fp32 maps: bit-exact.
  * bf16 mode (CF_BF16) is the throughput mode; its error is bf16 rounding of every stored
    activation (the reference itself run in bf16 shows head error mean 0.026).  It is checked
    against the bf16-EMULATING oracle (oracle/bf16_emulation.py, pinned to a hooked run of the reference) at
    one bf16 ulp + flip noise per kernel; the production instances and the engine layer by layer are in
    tests/test_bf16_parity.py.
"""
import os
import sys

import numpy as np
import pytest
import torch

import centerface_amd as cfa
from centerface_amd import ops
from oracle import bf16_emulation as E
from oracle import centerface_oracle as O

pytestmark = pytest.mark.gpu

F32 = dict(rtol=2e-5, atol=2e-5)
SPLIT = dict(rtol=1e-4, atol=1e-4)        # "fp32_split": fp32 storage, split-bf16 GEMM products (~2^-16 relative per product)
# the two modes that must meet north_star's 1e-3 against the reference: exact-fp32 MFMA and the split-bf16 tolerance mode
EXACT = ("fp32", "fp32_split")
BF16 = dict(rtol=2.5e-2, atol=2.5e-2)      # only for bf16 flavours of ops that are NOT on the engine's bf16 path (unfused stem, two-stage heads)


def _emu_close(got, ref, what, bf16_output=True):
    """bf16 kernels against the bf16-emulating oracle: one bf16 ulp + flip noise (oracle/bf16_emulation.py)."""
    ref = ref.numpy() if hasattr(ref, "numpy") else np.asarray(ref)
    r = np.abs(np.asarray(got, np.float64) - ref.astype(np.float64)) / E.tolerance(ref)
    stat = (float(r.max()), float((r > 1).mean()), float((r > 0.5).mean()), float((r > 0).mean()), int(r.size))
    assert E.accept(stat, bf16_output), (what, stat)


def _tol(dtype):
    return F32 if dtype == "fp32" else SPLIT if dtype == "fp32_split" else BF16


def _sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def test_library_is_the_gpu_path():
    import ctypes
    L = cfa._lib.lib()
    n = ctypes.c_int()
    assert L.cf_device_count(ctypes.byref(n)) == 0 and n.value >= 1
    assert L._name.endswith("libcenterface_hip.so")


# ------------------------------------------------------------------------------- per-op: goldens
@pytest.mark.parametrize("dtype", ["fp32", "fp32_split", "bf16"])
def test_conv_swish_flavours_vs_reference(golden, dtype):
    g = golden("ops")
    # cr0: stem 3->32 k3 s2 ; cr1..4: depthwise k3/k5 s1/s2 incl. borders, non-square ; cr5: 1x1
    y = ops.stem(g["cr0_x"], g["cr0_w"], dtype=dtype)
    np.testing.assert_allclose(y, g["cr0_y"], **_tol(dtype))
    for i in range(1, 5):
        cin, cout, k, s, groups = (int(v) for v in g["cr%d_cfg" % i])
        y = ops.conv_dw(g["cr%d_x" % i], g["cr%d_w" % i], k, s, dtype=dtype)
        if dtype == "bf16":
            _emu_close(y, E.dw_op(g["cr%d_x" % i], g["cr%d_w" % i], k, s), "cr%d" % i)
        else:
            np.testing.assert_allclose(y, g["cr%d_y" % i], **_tol(dtype), err_msg="cr%d" % i)
    y = ops.conv_pw(g["cr5_x"], g["cr5_w"], act="swish", dtype=dtype)
    if dtype == "bf16":
        _emu_close(y, E.pw_op(g["cr5_x"], g["cr5_w"], act="swish"), "cr5")
    else:
        np.testing.assert_allclose(y, g["cr5_y"], **_tol(dtype))


@pytest.mark.parametrize("k,s", [(3, 1), (5, 1), (3, 2), (5, 2)])
@pytest.mark.parametrize("C", [8, 16, 40])
def test_depthwise_narrow_channel_chunks_vs_oracle(C, k, s):
    """cf_dw.hip with channel chunks of ONE or two 16-byte groups (C = 8: cpp == 1 in bf16; C = 16 / 40: 2 / 5 groups, fp32: 2 / 4 / 10): round 6 found
    that `magic_div(1)` does not fit 32 bits -- every chunk fetched from pixel 0, wrong and fast -- in the strip kernel AND latent in round 1's kernel.
    Both dtypes, both kernels (stride 1: strip form, stride 2: one-vector form), maps that are not multiples of any tile, a batch."""
    rng = np.random.default_rng(C * 100 + k * 10 + s)
    x = rng.standard_normal((3, C, 37, 53)).astype(np.float32)
    w = (rng.standard_normal((C, 1, k, k)) * 0.3).astype(np.float32)
    ref = O.swish(torch.nn.functional.conv2d(torch.nn.functional.pad(torch.from_numpy(x), _same_pad(k, s)), torch.from_numpy(w), None, s, 0, 1, C)).numpy()
    y = ops.conv_dw(x, w, k, s, dtype="fp32")
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)
    _emu_close(ops.conv_dw(x, w, k, s, dtype="bf16"), E.dw_op(x, w, k, s), "dw C=%d k=%d s=%d" % (C, k, s))


@pytest.mark.parametrize("C,k,s,H", [(96, 3, 2, 320), (32, 3, 1, 320), (144, 3, 1, 160), (192, 5, 1, 80), (144, 5, 2, 160), (960, 3, 1, 20)])
def test_depthwise_production_shapes_vs_oracle(C, k, s, H):
    """The standalone depthwise at the network's own shapes (layer1.0 / 0.0 / 1.1 / 2.1 / 2.0 / 6.0 of a 640x640 input), a batch of three: several channel
    chunks per tile, hundreds of tiles, a workgroup count that is not a multiple of eight -- the XCD-contiguous work order (dw_xcd_remap), the incremental
    chunk addressing of the staging loop and the per-shape tile table, in both storage types."""
    rng = np.random.default_rng(C + 7 * k + s)
    x = rng.standard_normal((3, C, H, H)).astype(np.float32)
    w = (rng.standard_normal((C, 1, k, k)) * 0.3).astype(np.float32)
    ref = O.swish(torch.nn.functional.conv2d(torch.nn.functional.pad(torch.from_numpy(x), _same_pad(k, s)), torch.from_numpy(w), None, s, 0, 1, C)).numpy()
    np.testing.assert_allclose(ops.conv_dw(x, w, k, s, dtype="fp32"), ref, rtol=2e-5, atol=2e-5)
    _emu_close(ops.conv_dw(x, w, k, s, dtype="bf16"), E.dw_op(x, w, k, s), "dw C=%d k=%d s=%d H=%d" % (C, k, s, H))


def _same_pad(k, s):
    p = max(k - s, 0)
    return [p // 2, p - p // 2, p // 2, p - p // 2]


def _mbconv_gpu(x, sd, cin, cout, t, k, s, dtype):
    y, j = x, 0
    if t != 1:
        y = ops.conv_pw(y, sd["conv.0.1.weight"], act="swish", dtype=dtype)
        j = 1
    y = ops.conv_dw(y, sd["conv.%d.1.weight" % j], k, s, dtype=dtype)
    res = x if (cin == cout and s == 1) else None
    return ops.conv_pw(y, sd["conv.%d.weight" % (j + 1)], act="none", residual=res, dtype=dtype)


@pytest.mark.parametrize("dtype", ["fp32", "fp32_split", "bf16"])
def test_mbconv_blocks_vs_reference(golden, dtype):
    g = golden("ops")
    for i in range(9):
        cin, cout, t, k, s = (int(v) for v in g["mb%d_cfg" % i])
        sd = _sub(g, "mb%d_w_" % i)
        y = _mbconv_gpu(g["mb%d_x" % i], sd, cin, cout, t, k, s, dtype)
        if dtype in EXACT:
            np.testing.assert_allclose(y, g["mb%d_y" % i], rtol=1e-4, atol=1e-4, err_msg="mb%d" % i)
        else:       # three bf16 kernels: every intermediate is an HBM tensor
            j = 0 if t == 1 else 1
            emu = E.mbconv_unfused(E.q_bf16(torch.from_numpy(g["mb%d_x" % i])), sd["conv.0.1.weight"] if t != 1 else None,
                                   sd["conv.%d.1.weight" % j], sd["conv.%d.weight" % (j + 1)], k, s, cin == cout and s == 1)
            _emu_close(y, emu, "mb%d" % i)


@pytest.mark.parametrize("dtype", ["fp32", "fp32_split", "bf16"])
def test_fused_mbconv_vs_reference(golden, dtype):
    """The fused expand->dw->project kernel (cf_mbconv.hip) on every golden block shape it covers
    (small, non-square maps: every tile is an edge tile)."""
    g = golden("ops")
    done = 0
    for i in range(9):
        cin, cout, t, k, s = (int(v) for v in g["mb%d_cfg" % i])
        if t == 1 or cout > 96:
            continue
        sd = _sub(g, "mb%d_w_" % i)
        y = ops.mbconv(g["mb%d_x" % i], sd["conv.0.1.weight"], sd["conv.1.1.weight"], sd["conv.2.weight"], k, s, dtype=dtype)
        if dtype in EXACT:
            np.testing.assert_allclose(y, g["mb%d_y" % i], rtol=1e-4, atol=1e-4, err_msg="mb%d" % i)
        else:
            we, wp = sd["conv.0.1.weight"], sd["conv.2.weight"]
            emu = E.mbconv_fused(E.q_bf16(torch.from_numpy(g["mb%d_x" % i])), we.reshape(we.shape[0], -1), sd["conv.1.1.weight"],
                                 wp.reshape(wp.shape[0], -1), k, s, cin == cout and s == 1)
            _emu_close(y, emu, "mb%d" % i)
        done += 1
    assert done == 6


@pytest.mark.parametrize("cfg", [(16, 24, 3, 2, 70, 90), (24, 24, 3, 1, 41, 37), (24, 32, 5, 2, 64, 48),
                                 (32, 32, 5, 1, 33, 50), (32, 64, 3, 2, 45, 38), (64, 64, 3, 1, 21, 30),
                                 (64, 96, 5, 1, 40, 40), (96, 96, 5, 1, 24, 17)])
def test_fused_mbconv_multi_tile_vs_oracle(cfg):
    """Interior + edge tiles, odd sizes, batch > 1, against the oracle's MBConv restatement (fp32)."""
    cin, cout, k, s, H, W = cfg
    rng = np.random.default_rng(cin * 1000 + k * 10 + s)
    hid = cin * 6
    sd = {"b.conv.0.1.weight": (rng.standard_normal((hid, cin, 1, 1)) * 1.5 / np.sqrt(cin)).astype(np.float32),
          "b.conv.1.1.weight": (rng.standard_normal((hid, 1, k, k)) * 1.5 / k).astype(np.float32),
          "b.conv.2.weight": (rng.standard_normal((cout, hid, 1, 1)) / np.sqrt(hid)).astype(np.float32)}
    x = rng.standard_normal((2, cin, H, W)).astype(np.float32)
    ref = O.mbconv(torch.from_numpy(x), {k_: torch.from_numpy(v) for k_, v in sd.items()}, "b", cin, cout, 6, k, s).numpy()
    for dt in EXACT:
        y = ops.mbconv(x, sd["b.conv.0.1.weight"], sd["b.conv.1.1.weight"], sd["b.conv.2.weight"], k, s, dtype=dt)
        np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4, err_msg=dt)
    yb = ops.mbconv(x, sd["b.conv.0.1.weight"], sd["b.conv.1.1.weight"], sd["b.conv.2.weight"], k, s, dtype="bf16")
    emu = E.mbconv_fused(E.q_bf16(torch.from_numpy(x)), sd["b.conv.0.1.weight"].reshape(hid, cin), sd["b.conv.1.1.weight"],
                         sd["b.conv.2.weight"].reshape(cout, hid), k, s, cin == cout and s == 1)
    _emu_close(yb, emu, cfg)


def test_fp32_second_generation_kernel_on_every_block_shape():
    """``mbconv_f32_kernel`` (cf_mbconv4.hip: SGPR taps, permlane32 swap into the project MFMA) runs layer2.1 by default;
    ``CF_F4_VARIANT=1`` puts all eight backbone block shapes on it.  The variant is read once per process, so the multi-tile
    oracle comparison above is re-run in a child process with it set -- the kernel the table does not pick must stay correct."""
    import subprocess
    env = dict(os.environ, CF_F4_VARIANT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_fused_mbconv_multi_tile_vs_oracle or test_network_fp32_vs_reference_goldens"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "10 passed" in r.stdout, r.stdout[-1000:]        # 8 block shapes (fp32 + split mode each) + the network goldens in both modes


@pytest.mark.parametrize("dtype", ["fp32", "fp32_split", "bf16"])
def test_conv1x1bn_idaup_heads_vs_reference(golden, dtype):
    g = golden("ops")
    sd = _sub(g, "c1bn_w_")
    sc = sd["conv_last.1.weight"].astype(np.float64) / np.sqrt(sd["conv_last.1.running_var"].astype(np.float64) + 1e-5)
    w = (sd["conv_last.0.weight"].reshape(24, 320) * sc[:, None]).astype(np.float32)
    b = (sd["conv_last.1.bias"] - sd["conv_last.1.running_mean"] * sc).astype(np.float32)
    y = ops.conv_pw(g["c1bn_x"], w, act="swish", bias=b, dtype=dtype)
    if dtype in EXACT:
        np.testing.assert_allclose(y, g["c1bn_y"], rtol=1e-4, atol=1e-4)
    else:
        _emu_close(y, E.conv_last(E.q_bf16(torch.from_numpy(g["c1bn_x"])), O.to_torch_sd(sd)), "conv_1x1_bn")
    for i in range(3):
        y = ops.idaup(g["ida%d_lo" % i], g["ida%d_skip" % i], _sub(g, "ida%d_w_" % i), "up", dtype=dtype)
        if dtype in EXACT:
            np.testing.assert_allclose(y, g["ida%d_y" % i], rtol=1e-4, atol=1e-4)
        else:
            _emu_close(y, E.idaup(E.q_bf16(torch.from_numpy(g["ida%d_lo" % i])), E.q_bf16(torch.from_numpy(g["ida%d_skip" % i])),
                                  O.to_torch_sd(_sub(g, "ida%d_w_" % i)), "up"), "ida%d" % i)
    full = cfa.weights.synthetic_state_dict(0)
    for collapse in (False, True):
        out = ops.heads(g["head_x"], full, collapse=collapse, dtype=dtype)
        if dtype == "bf16" and collapse:            # the engine's bf16 flavour: collapsed 3x3 24 -> 15, fp32 out
            emu = E.heads(E.q_bf16(torch.from_numpy(g["head_x"])), O.to_torch_sd(full))
            for k in ("hm", "wh", "lm", "reg"):
                _emu_close(out[k], emu[k], "head." + k, bf16_output=False)
        else:
            np.testing.assert_allclose(out["lm"], g["head_y"], **(dict(rtol=1e-4, atol=1e-4) if dtype in EXACT else BF16))


@pytest.mark.parametrize("dtype", ["fp32", "fp32_split", "bf16"])
def test_shufflev2_block_vs_reference(golden, dtype):
    """ShuffleV2Block (model/blocks.py:4-62) through ONE product entry point (cf_op_shufflev2): BN fold in the
    runtime, channel shuffle and concat as channel addressing inside the kernels -- against the reference's goldens
    (stride 1 / 2, 3x3 / 5x5; fp32) and the bf16-emulating oracle (bf16)."""
    g = golden("ops")
    for i in range(4):
        inp, oup, mid, k, s = (int(v) for v in g["sh%d_cfg" % i])
        sd, x = _sub(g, "sh%d_w_" % i), g["sh%d_x" % i]
        y = ops.shuffle_v2_block(x, sd, inp, oup, mid, k, s, dtype=dtype)
        assert y.shape == g["sh%d_y" % i].shape
        if dtype in EXACT:
            np.testing.assert_allclose(y, g["sh%d_y" % i], rtol=1e-4, atol=1e-4, err_msg="sh%d" % i)
            if s == 1:                                   # the pass-through half is a copy: bit-exact
                assert np.array_equal(y[:, :inp], x[:, 0::2])
        else:
            _emu_close(y, E.shuffle_v2_block(x, sd, inp, oup, mid, k, s), "sh%d" % i)
    with pytest.raises(ValueError):
        ops.shuffle_v2_block(g["sh0_x"], _sub(g, "sh0_w_"), 24, 48, 20, 3, 1)         # mid not a multiple of 8: loud


# ------------------------------------------------------------------------------- whole network
@pytest.mark.parametrize("dtype", EXACT)
def test_network_fp32_vs_reference_goldens(golden, dtype):
    g = golden("net")
    sd = cfa.weights.synthetic_state_dict(0)
    assert cfa.weights.fingerprint(sd) == str(g["weights_fingerprint"])
    for tag in "abc":
        x = g["x_" + tag]
        eng = cfa.Engine(x.shape[2], x.shape[3], max_batch=x.shape[0], dtype=dtype, weights=sd)
        out = eng.forward(x)
        for h in ("hm", "wh", "lm", "reg"):
            np.testing.assert_allclose(out[h], g["%s_%s" % (h, tag)], rtol=1e-3, atol=1e-3, err_msg=h + tag)
        eng.close()


@pytest.mark.parametrize("dtype", EXACT)
def test_network_unfused_path_fp32(golden, dtype):
    """CF_FLAG_NO_FUSE: the three-kernel MBConv path stays parity-green too."""
    g = golden("net")
    x = g["x_b"]
    eng = cfa.Engine(64, 96, max_batch=2, dtype=dtype, fuse=False)
    out = eng.forward(x)
    for h in ("hm", "wh", "lm", "reg"):
        np.testing.assert_allclose(out[h], g["%s_b" % h], rtol=1e-3, atol=1e-3)
    eng.close()


@pytest.mark.parametrize("dtype", EXACT)
@pytest.mark.parametrize("collapse", [False, True])
def test_network_head_flavours_fp32(golden, collapse, dtype):
    """Both head flavours of the fp32 parity mode against the reference goldens at 1e-3: the two-stage kernel
    (conv3x3 + b -> conv1x1 + b in the reference's operation order, collapse_heads=False) and the default collapsed
    one (the pair is linear -- model/centernet.py:249-256 has nothing between the convs -- so folding it in float64
    into one 3x3 24->15 conv is exact algebra and 6x fewer flops); they agree with each other to fp32 rounding."""
    g = golden("net")
    x = g["x_b"]
    eng = cfa.Engine(64, 96, max_batch=2, dtype=dtype, collapse_heads=collapse)
    out = eng.forward(x)
    other = cfa.Engine(64, 96, max_batch=2, dtype=dtype, collapse_heads=not collapse)
    out2 = other.forward(x)
    for h in ("hm", "wh", "lm", "reg"):
        np.testing.assert_allclose(out[h], g["%s_b" % h], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(out[h], out2[h], **(F32 if dtype == "fp32" else SPLIT))
    eng.close(); other.close()


@pytest.mark.parametrize("dtype", EXACT)
def test_uint8_image_path_fp32(golden, dtype):
    """uint8 BGR image -> fused normalisation (centerface.py:32-37) -> net -> sigmoid/clamp (:43)."""
    g = golden("net")
    eng = cfa.Engine(64, 96, max_batch=1, dtype=dtype)
    eng.forward_enqueue(g["img_u8"][None])
    out = eng.heads(sigmoid_hm=True)
    np.testing.assert_allclose(out["hm_sigmoid"], g["img_hm_sigmoid"], rtol=1e-3, atol=1e-3)
    for h in ("wh", "lm", "reg"):
        np.testing.assert_allclose(out[h], g["img_" + h], rtol=1e-3, atol=1e-3)
    eng.close()


def test_network_bf16_vs_hooked_reference(golden):
    """bf16 engine against the REFERENCE run with the engine's roundings inserted by hooks (net_bf16emu.npz):
    the 32x32 case end to end (no rounding flip on so small a map: agreement to 2e-3 of a map whose rms is 1.4-4.3),
    the larger cases block by block on the golden's... no: on the ENGINE's own block inputs (teacher forcing, see
    tests/test_bf16_parity.py), with the golden as the independent end-to-end cross-check at drift level."""
    g = golden("net_bf16emu")
    sd = cfa.weights.synthetic_state_dict(0)
    eng = cfa.Engine(32, 32, max_batch=1, dtype="bf16")
    out = eng.forward(g["x_a"])
    for h in ("hm", "wh", "lm", "reg"):
        np.testing.assert_allclose(out[h], g[h + "_a"], rtol=0, atol=2e-3, err_msg=h)
    eng.close()
    for tag in "bc":
        x = g["x_" + tag]
        eng = cfa.Engine(x.shape[2], x.shape[3], max_batch=1, dtype="bf16")
        out = eng.forward(x)
        for h in ("hm", "wh", "lm", "reg"):
            ref = g["%s_%s" % (h, tag)]
            d = np.abs(out[h] - ref)
            rms = float(np.sqrt((ref ** 2).mean()))
            assert d.mean() <= 0.006 * rms and d.max() <= 0.12 * rms, (tag, h, float(d.mean()) / rms, float(d.max()) / rms)
        # first block: bit-level agreement with the hooked reference (nothing upstream to drift)
        _emu_close(eng.trace(x, 0), E.from_bf16_bits(g["layer0.0_" + tag]), "layer0.0_" + tag)
        eng.close()


def test_forward_errors_are_loud():
    eng = cfa.Engine(32, 32, max_batch=1)
    with pytest.raises(ValueError):
        eng.forward_enqueue(np.zeros((2, 3, 32, 32), np.float32))      # B > max_batch
    with pytest.raises(ValueError):
        cfa.Engine(100, 100)                                            # not a multiple of 32
    sd = cfa.weights.synthetic_state_dict(0)
    bad = dict(sd); bad.pop("hm.1.bias")
    with pytest.raises(ValueError):
        eng.load_state_dict(bad)
    eng.close()


# ------------------------------------------------------------------------------- decode D3
def test_ctdet_decode_bit_exact_vs_reference(golden):
    g = golden("decode_d3")
    for tag in "smlx":
        heat, wh, reg, K = g[tag + "_heat"], g[tag + "_wh"], g[tag + "_reg"], int(g[tag + "_K"])
        pos = g[tag + "_topk_score"] > 0          # strictly distinct region (zeros tie, order unspecified in torch)
        dets, _, inds = ops.ctdet_decode(heat, wh, reg, K)
        assert np.array_equal(inds[pos], g[tag + "_topk_inds"][pos]), tag
        assert np.array_equal(dets[pos], g[tag + "_det"][pos]), tag
        assert np.array_equal(dets[..., 4], g[tag + "_topk_score"]), tag
        dets, _, _ = ops.ctdet_decode(heat, wh, None, K)
        assert np.array_equal(dets[pos], g[tag + "_det_noreg"][pos]), tag
        # full agreement with the oracle's tie rule (lower index first), landmarks included
        lm = np.random.default_rng(3).standard_normal((heat.shape[0], 10) + heat.shape[2:]).astype(np.float32)
        d, l, i = ops.ctdet_decode(heat, wh, reg, K, lm)
        od, ol, oi = O.ctdet_decode(heat, wh, reg, K, lm)
        assert np.array_equal(i, oi) and np.array_equal(d, od) and np.array_equal(l, ol), tag


def test_ctdet_decode_ties_and_plateaus(golden):
    g = golden("decode_d3")
    heat = g["tie_heat"]
    wh = np.ones((1, 2, 6, 6), np.float32)
    d, _, i = ops.ctdet_decode(heat, wh, None, 8)
    od, _, oi = O.ctdet_decode(heat, wh, None, 8)
    assert np.array_equal(i, oi) and np.array_equal(d, od)
    assert list(i[0, :3]) == [4 * 6 + 5, 2 * 6 + 2, 2 * 6 + 3]       # 0.9, then the 0.7 plateau by index
    # constant map: every cell is a plateau peak -> indices 0..K-1
    heat = np.full((2, 1, 8, 8), 0.5, np.float32)
    d, _, i = ops.ctdet_decode(heat, np.ones((2, 2, 8, 8), np.float32), None, 64)
    assert np.array_equal(i, np.tile(np.arange(64), (2, 1)))
    # K == H*W, negative and zero scores, -0.0
    rng = np.random.default_rng(0)
    heat = rng.standard_normal((1, 1, 5, 7)).astype(np.float32)
    heat[0, 0, 0, 0] = -0.0
    d, _, i = ops.ctdet_decode(heat, np.ones((1, 2, 5, 7), np.float32), None, 35)
    od, _, oi = O.ctdet_decode(heat, np.ones((1, 2, 5, 7), np.float32), None, 35)
    assert np.array_equal(i, oi) and np.array_equal(d, od)
    with pytest.raises(ValueError):
        ops.ctdet_decode(heat, np.ones((1, 2, 5, 7), np.float32), None, 36)     # K > H*W


def test_ctdet_decode_large_crowd_map():
    """Config 5 shape: 320x320 map, K=1000, B=4 -- against the oracle (numpy stable argsort)."""
    rng = np.random.default_rng(11)
    B, H, W, K = 4, 320, 320, 1000
    heat = rng.uniform(1e-4, 0.9999, (B, 1, H, W)).astype(np.float32)
    heat[:, :, ::7, ::5] = 0.25                     # many exact ties
    wh = rng.uniform(1, 9, (B, 2, H, W)).astype(np.float32)
    reg = rng.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    lm = rng.standard_normal((B, 10, H, W)).astype(np.float32)
    d, l, i = ops.ctdet_decode(heat, wh, reg, K, lm)
    od, ol, oi = O.ctdet_decode(heat, wh, reg, K, lm)
    assert np.array_equal(i, oi) and np.array_equal(d, od) and np.array_equal(l, ol)
    assert (np.diff(d[..., 4], axis=1) <= 0).all()


@pytest.mark.parametrize("case", ["k_equals_hw", "constant_map", "negative_scores", "zeros_fill", "k2000_big_path",
                                  "map_over_2p17_cells", "map_over_2p17_k3000", "sparse_positive"])
def test_ctdet_decode_no_limits_and_zero_fill(case):
    """The collect + select decode beyond the one-kernel version's limits (K <= 1024, h*w <= 2^17) and through its
    rare branches, bit-exact against the oracle's ctdet_decode (centerface_ext.py:11-82): K = h*w, a constant map
    (every cell is a peak: the list is the whole map), negative scores (they rank BELOW the +0 of suppressed cells),
    fewer positive peaks than K (the lowest-index suppressed cells fill in, in index order), K > 1024 (global-memory
    sort), maps of more than 2^17 cells."""
    rng = np.random.default_rng(sum(ord(c) for c in case))
    B, h, w, K = 2, 24, 40, 100
    heat = rng.uniform(1e-4, 0.9999, (B, 1, h, w)).astype(np.float32)
    if case == "k_equals_hw":
        K = h * w
    elif case == "constant_map":
        heat[:] = 0.25; h, w = 80, 96; heat = np.full((B, 1, h, w), 0.25, np.float32); K = 300
    elif case == "negative_scores":
        heat = rng.standard_normal((B, 1, h, w)).astype(np.float32); K = 700
    elif case == "zeros_fill":
        heat = np.zeros((B, 1, h, w), np.float32)
        for b in range(B):
            for _ in range(30):
                heat[b, 0, rng.integers(h), rng.integers(w)] = rng.uniform(0.3, 0.9)
        K = 200
    elif case == "k2000_big_path":
        h, w, K = 96, 96, 2000
        heat = rng.uniform(1e-4, 0.9999, (B, 1, h, w)).astype(np.float32)
    elif case == "map_over_2p17_cells":
        B, h, w, K = 1, 400, 400, 100
        heat = rng.uniform(1e-4, 0.9999, (B, 1, h, w)).astype(np.float32)
    elif case == "map_over_2p17_k3000":
        B, h, w, K = 1, 384, 512, 3000
        heat = rng.uniform(1e-4, 0.9999, (B, 1, h, w)).astype(np.float32)
    elif case == "sparse_positive":
        heat = (rng.uniform(0, 1, (B, 1, h, w)) > 0.97).astype(np.float32) * rng.uniform(0.2, 0.9, (B, 1, h, w)).astype(np.float32)
        heat -= 0.05 * (rng.uniform(0, 1, (B, 1, h, w)) > 0.99)                      # a few negative cells as well
        heat = heat.astype(np.float32); K = 960
    wh = rng.uniform(1, 20, (B, 2, h, w)).astype(np.float32)
    reg = rng.uniform(0, 1, (B, 2, h, w)).astype(np.float32)
    lm = rng.standard_normal((B, 10, h, w)).astype(np.float32)
    dets, lms, inds = ops.ctdet_decode(heat, wh, reg, K, lm)
    rd, rl, ri = O.ctdet_decode(heat, wh, reg, K, lm)
    assert np.array_equal(inds, ri), case
    assert np.array_equal(dets, rd) and np.array_equal(lms, rl), case
    with pytest.raises(ValueError):
        ops.ctdet_decode(heat, wh, reg, h * w + 1, lm)


def test_ctdet_post_process_vs_oracle():
    """utils/post_process.py:83-100 on the GPU: standalone on explicit dets and fused into the top-K
    decode epilogue, against the oracle restatement (float64 matrices agree to rounding)."""
    rng = np.random.default_rng(8)
    B, K = 3, 50
    dets = np.zeros((B, K, 6), np.float32)
    dets[:, :, :4] = rng.uniform(-5, 170, (B, K, 4)); dets[:, :, 4] = rng.uniform(0, 1, (B, K))
    c = np.array([[320, 320], [360, 239], [100.5, 77.25]], np.float32)
    s = np.array([640.0, 736.0, 512.0], np.float32)
    ref_in = dets.copy()
    ref = O.ctdet_post_process(ref_in, c, s, 160, 160, 1)
    got_in = dets.copy()
    got = cfa.post_process.ctdet_post_process(got_in, c, s, 160, 160, 1)
    np.testing.assert_allclose(got_in, ref_in, rtol=1e-6, atol=1e-4)
    assert [sorted(g) for g in got] == [sorted(r) for r in ref] and len(got[0][1]) == K
    np.testing.assert_allclose(np.asarray(got[2][1]), np.asarray(ref[2][1]), rtol=1e-6, atol=1e-4)
    t = cfa.post_process.get_affine_transform(c[1], s[1], 0, (160, 160), inv=1)
    np.testing.assert_allclose(t, O.get_affine_transform(c[1], s[1], 0, (160, 160), inv=1), rtol=1e-9, atol=1e-9)
    pts = rng.uniform(0, 160, (7, 2)).astype(np.float32)
    np.testing.assert_allclose(cfa.post_process.transform_preds(pts, c[2], s[2], (160, 160)),
                               O.transform_preds(pts, c[2], s[2], (160, 160)), rtol=1e-6, atol=1e-4)
    # fused into the decode kernel
    eng = cfa.Engine(64, 96, max_batch=2, dtype="fp32")
    eng.forward_enqueue(rng.integers(0, 256, (2, 64, 96, 3), dtype=np.uint8))
    plain, _, inds = eng.decode_topk(K=20)
    c2, s2 = np.array([[48, 32], [50, 30]], np.float32), np.array([96.0, 120.0], np.float32)
    fused, _, inds2 = eng.decode_topk(K=20, post=(c2, s2))
    assert np.array_equal(inds, inds2)
    want = plain.copy()
    O.ctdet_post_process(want, c2, s2, eng.h, eng.w, 1)
    np.testing.assert_allclose(fused, want, rtol=1e-6, atol=1e-4)
    eng.close()


# ------------------------------------------------------------------------------- decode D1 + API
def test_decode_d1_and_nms_bit_exact_vs_reference(golden):
    g = golden("decode_d1")
    face = cfa.CenterFace(32, 32)
    for tag in "ab":
        b, l = face.decode(g[tag + "_hm"], g[tag + "_wh"], g[tag + "_off"], g[tag + "_lm"],
                           tuple(int(v) for v in g[tag + "_size"]), threshold=0.9)
        assert np.array_equal(b, g[tag + "_boxes"]), tag
        assert np.array_equal(l, g[tag + "_lms"]), tag
    b, l = face.decode(np.full((1, 1, 8, 8), 0.2, np.float32), np.ones((1, 2, 8, 8), np.float32),
                       np.zeros((1, 2, 8, 8), np.float32), np.zeros((1, 10, 8, 8), np.float32), (32, 32))
    assert b == [] and l == []
    for thr in (0.3, 0.5):
        keep = face.nms(g["nms_boxes"], g["nms_scores"], thr)
        assert keep == [int(v) for v in g["nms_keep_%d" % int(thr * 10)]]
    for (h, w), ref in zip(g["tf_in"], g["tf_out"]):
        assert np.array_equal(np.asarray(face.transform(int(h), int(w)), np.float64), ref)


def test_decode_d2_and_get_detections(golden):
    """eval_widerface.decode / get_detections (eval_widerface.py:76-152) on the GPU path: D2 decode
    bit-exact against the reference goldens, batched get_detections against the oracle."""
    from centerface_amd import eval_widerface as ew
    g = golden("decode_d2")
    for tag in "abc":
        h, w = g[tag + "_hm"].shape[1:]
        b = ew.decode(g[tag + "_hm"], g[tag + "_wh"], g[tag + "_off"], None, (h * 4, w * 4), threshold=float(g[tag + "_thr"]))
        assert np.array_equal(np.asarray(b, np.float32).reshape(-1, 5), g[tag + "_boxes"].reshape(-1, 5)), tag
    assert ew.decode(np.full((1, 4, 4), 0.1, np.float32), np.ones((2, 4, 4), np.float32), np.zeros((2, 4, 4), np.float32),
                     None, (16, 16), threshold=0.5) == []
    assert ew.nms(g["a_boxes"][:, :4], g["a_boxes"][:, 4], 0.3) == O.nms_greedy(g["a_boxes"][:, :4], g["a_boxes"][:, 4], 0.3)
    # batched harness: forward on the GPU, then D2 of OUR heads must equal the oracle's D2 of the same heads
    rng = np.random.default_rng(31)
    x = rng.standard_normal((3, 3, 64, 96)).astype(np.float32)
    eng = cfa.Engine(64, 96, max_batch=2, dtype="fp32")
    dets = ew.get_detections({"input": x}, eng, threshold=0.2)                  # clamps to the reference's hard-coded (640, 640), :88
    dets_own = ew.get_detections({"input": x}, eng, threshold=0.2, size=None)   # clamps to the engine's input size
    assert len(dets) == 3 and len(dets_own) == 3
    for i in range(3):
        eng.forward_enqueue(x[i:i + 1])
        hd = eng.heads(sigmoid_hm=True)
        for got, size in ((dets[i], (640, 640)), (dets_own[i], (64, 96))):
            ref = O.decode_d2(hd["hm_sigmoid"][0], hd["wh"][0], hd["reg"][0], size, threshold=0.2)
            assert np.array_equal(np.asarray(got, np.float32).reshape(-1, 5), np.asarray(ref, np.float32).reshape(-1, 5))
    eng.close()


def test_bbox_overlap_and_evaluate_on_the_device(golden):
    """SURVEY 8f N2, second half: eval_widerface.bbox_overlap / evaluate through cf_op_box_match -- the IoU matrix bit-exact
    against the reference goldens, evaluate's (recall, precision) equal to the reference's to the last bit at both thresholds,
    then the whole evaluate loop (forward + D2 decode + NMS + match, all on the device) against the oracle's bookkeeping on
    the engine's own detections, and box_match of a detection set against itself."""
    from centerface_amd import eval_widerface as ew
    g = golden("eval_metrics")
    for i in g["ov_cases"]:
        ov = ew.bbox_overlap(g["ov%d_boxes" % i], g["ov%d_query" % i])
        assert ov.dtype == np.float64 and np.array_equal(ov, g["ov%d_out" % i]), i
    picked, annots = [], []
    for bi, nimg in enumerate(g["eval_batches"]):
        picked.append([g["eval_b%d_i%d_det" % (bi, j)] if len(g["eval_b%d_i%d_det" % (bi, j)]) else [] for j in range(int(nimg))])
        annots.append([g["eval_b%d_i%d_gt" % (bi, j)] for j in range(int(nimg))])
    val = [{"meta": {"gt_det": a}, "_picked": p} for a, p in zip(annots, picked)]
    for thr, key in ((0.5, "eval_thr50"), (0.35, "eval_thr35")):
        r, p = ew.evaluate(val, None, threshold=thr, detections=lambda data, model: data["_picked"])
        assert (r, p) == tuple(g[key]), (thr, r, p)
    # random sets vs the oracle (sizes beyond one workgroup pass, sub-pixel boxes, a 5- and an 8-float row stride)
    rng = np.random.default_rng(9)
    for n, k in ((300, 257), (1, 600), (513, 2)):
        b = rng.uniform(0, 600, (n, 2)).astype(np.float32); b = np.concatenate([b, b + rng.uniform(1, 90, (n, 2)).astype(np.float32), rng.uniform(0, 1, (n, 1)).astype(np.float32)], 1)
        q = rng.uniform(0, 600, (k, 2)).astype(np.float32); q = np.concatenate([q, q + rng.uniform(1, 90, (k, 2)).astype(np.float32), np.zeros((k, 4), np.float32)], 1)
        assert np.array_equal(ew.bbox_overlap(b, q), O.bbox_overlap(b[:, :4], q[:, :4]))
        for thr in (0.5, 0.1):
            assert tuple(ew.match_counts([b], [q], thr)[0]) == O.evaluate_counts(b, q, thr)
    # the whole loop on the device: synthetic annotations = jittered detections of the engine itself.  The reference's decoders
    # take the box size linearly from the wh head (eval_widerface.py:100), so the synthetic weights get a positive size bias
    S = 96
    sd = dict(cfa.weights.synthetic_state_dict(0))
    sd["wh.1.bias"] = (sd["wh.1.bias"] + 6.0).astype(np.float32)
    eng = cfa.Engine(S, S, max_batch=4, dtype="fp32", weights=sd)
    batches = []
    for bi in range(2):
        x = rng.standard_normal((4, 3, S, S)).astype(np.float32)
        dets = ew.get_detections({"input": x}, eng, threshold=0.3)
        gts = []
        for d in dets:
            gt = np.full((16, 4), -1.0, np.float32)
            if len(d):
                m = min(len(d), 12)
                gt[:m] = d[:m, :4] + rng.normal(0, 1.5, (m, 4)).astype(np.float32)
            gts.append(gt)
        batches.append({"input": x, "meta": {"gt_det": gts}})
    r, p = ew.evaluate(batches, eng, threshold=0.5, detections=lambda data, model: ew.get_detections(data, model, threshold=0.3))
    picked = [ew.get_detections(b, eng, threshold=0.3) for b in batches]
    assert (r, p) == O.evaluate(picked, [b["meta"]["gt_det"] for b in batches], threshold=0.5)
    assert 0.0 < r <= 1.0 and 0.0 < p <= 1.0
    nonempty = [d for d in picked[0] if len(d)]
    bm = ew.box_match(nonempty, nonempty)
    assert bm["recall"] == 1.0 and bm["precision"] == 1.0 and bm["images"] == len(nonempty)
    eng.close()


def test_device_resize_and_non32_sizes(tmp_path):
    """centerface.py:30 on the device: cv2.resize's fixed-point INTER_LINEAR (integer arithmetic: BIT-EXACT against
    the oracle's restatement of OpenCV's published algorithm; parity with a cv2 binary is unpinned), up- and
    down-scaling, then the full __call__ on a 478x720 image (imgs/1.jpg's size, BASELINE configs[0] geometry) --
    from a JPEG file through the PIL loader as well, and detections equal the oracle's detect() on the same pixels."""
    rng = np.random.default_rng(12)
    sd = cfa.weights.synthetic_state_dict(0)
    for (h, w) in ((50, 70), (33, 97), (478, 720)):
        img = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
        face = cfa.CenterFace(h, w, weights=sd, max_batch=2)
        assert (face.img_h_new, face.img_w_new) == (O.transform(h, w)[:2])
        face.engine.forward_resized_enqueue(img)
        got = face.engine.resized_input()
        want = np.stack([O.resize_bilinear_u8(im, face.img_h_new, face.img_w_new) for im in img])
        assert np.array_equal(got, want), (h, w)
        if (h, w) != (478, 720):
            face.close()
    # a 2.3x DOWN-scale through the same kernel (any source size -> the context's size)
    big = rng.integers(0, 256, (1, 150, 223, 3), dtype=np.uint8)
    small = cfa.Engine(64, 96, max_batch=1, weights=sd)
    small.forward_resized_enqueue(big)
    assert np.array_equal(small.resized_input()[0], O.resize_bilinear_u8(big[0], 64, 96))
    small.close()
    res = face.detect_batch(list(img))
    assert len(res) == 2
    for b in range(2):
        rd, rl = O.detect(O.to_torch_sd(sd), img[b])
        dets, lms = res[b]
        assert dets.shape == rd.shape and lms.shape == rl.shape
        # same pixels in, fp32 engine within 1e-3 of the oracle's heads: same candidates up to threshold-crossers
        assert np.abs(dets - rd).max() <= 1.0 if len(rd) else True
    # configs[0]: a 720x478 JPEG from disk (PIL decode -> BGR) through CenterFace.__call__
    from PIL import Image
    from centerface_amd import demo
    path = str(tmp_path / "1.jpg")
    Image.fromarray(img[0][:, :, ::-1]).save(path, quality=95)
    frame = demo.imread(path)
    assert frame.shape == (478, 720, 3) and frame.dtype == np.uint8
    assert np.array_equal(frame, np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    dets, lms = demo.detect_file(face, path)
    d2, l2 = face(frame)
    assert np.array_equal(dets, d2) and np.array_equal(lms, l2)
    face.close()


def test_centerface_call_matches_oracle():
    """CenterFace.__call__ end to end (identity-resize sizes) vs the oracle's restatement of it."""
    rng = np.random.default_rng(21)
    sd = cfa.weights.synthetic_state_dict(0)
    tsd = O.to_torch_sd(sd)
    for (H, W) in ((64, 96), (128, 128)):
        face = cfa.CenterFace(H, W, weights=sd, dtype="fp32")
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        dets, lms = face(img, threshold=0.35)
        rd, rl = O.detect(tsd, img)
        assert dets.dtype == np.float32 and dets.shape[1] == 5 and lms.shape[1] == 10
        # candidate sets can differ only for cells within 1e-3 of the 0.3 threshold or IoU ties;
        # require the same count and element-wise agreement after the floor rescale
        assert dets.shape == rd.shape, (dets.shape, rd.shape)
        np.testing.assert_allclose(dets[:, 4], rd[:, 4], atol=1e-3)
        assert (np.abs(dets[:, :4] - rd[:, :4]) <= 1.0).all()        # floor() discontinuity
        assert (np.abs(dets[:, :4] - rd[:, :4]) > 0).mean() < 0.05
        assert (np.abs(lms - rl) <= 1.0).all()
    # empty result keeps the reference's shapes (centerface.py:59-62)
    sd2 = dict(sd); sd2["hm.1.bias"] = np.full((1,), -30.0, np.float32)
    face = cfa.CenterFace(64, 64, weights=sd2)
    dets, lms = face(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8))
    assert dets.shape == (0, 5) and lms.shape == (0, 10) and dets.dtype == np.float32


@pytest.mark.parametrize("hw,mb", [((128, 160), 1), ((128, 160), 3), ((97, 131), 2)])
def test_detect_stream_equals_call(hw, mb):
    """CenterFace.detect_stream (two contexts, the next chunk's forward enqueued before the previous decode is collected):
    the same results, in order, as one __call__ per image -- identity-size and device-resized inputs, chunked and not,
    a count that does not fill the last chunk, and a second pass over the same object."""
    rng = np.random.default_rng(hw[0] + mb)
    face = cfa.CenterFace(hw[0], hw[1], dtype="bf16", max_batch=mb)
    imgs = [rng.integers(0, 256, hw + (3,), dtype=np.uint8) for _ in range(7)]
    want = [face(im) for im in imgs]
    for _ in range(2):
        got = list(face.detect_stream(iter(imgs)))
        assert len(got) == len(want)
        for (d, l), (wd, wl) in zip(got, want):
            assert np.array_equal(d, wd) and np.array_equal(l, wl)
    assert list(face.detect_stream([])) == []
    face.close()


# ------------------------------------------------------------------------------- full size
@pytest.mark.parametrize("dtype", EXACT)
def test_full_size_640_fp32_vs_oracle_topk(dtype):
    """BASELINE config 2 geometry at small batch: 640x640, fp32 parity mode, heads within 1e-3 of the
    oracle and top-100 indices identical wherever the oracle's score gaps exceed the head error."""
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (2, 640, 640, 3), dtype=np.uint8)
    sd = cfa.weights.synthetic_state_dict(0)
    eng = cfa.Engine(640, 640, max_batch=2, dtype=dtype, weights=sd)
    eng.forward_enqueue(img)
    got = eng.heads(sigmoid_hm=True)
    x = np.concatenate([O.preprocess(im) for im in img])
    ref = O.forward(O.to_torch_sd(sd), torch.from_numpy(x))
    for k in ("hm", "wh", "lm", "reg"):
        np.testing.assert_allclose(got[k], ref[k].numpy(), rtol=1e-3, atol=1e-3, err_msg=k)
    ref_hm = O.sigmoid_clamp(ref["hm"]).numpy()
    np.testing.assert_allclose(got["hm_sigmoid"], ref_hm, atol=1e-3)
    dets, lms, inds = eng.decode_topk(K=100)
    rdet, _, rinds = O.ctdet_decode(ref_hm, ref["wh"].numpy(), ref["reg"].numpy(), 100, ref["lm"].numpy())
    # decode of OUR heads is bit-exact against the oracle decode of the same heads ...
    odet, olms, oinds = O.ctdet_decode(got["hm_sigmoid"], got["wh"], got["reg"], 100, got["lm"])
    assert np.array_equal(inds, oinds) and np.array_equal(dets, odet) and np.array_equal(lms, olms)
    # ... and against the oracle's own heads the ranked indices agree wherever the reference's
    # neighbouring scores are separated by more than the measured head error
    err = float(np.abs(got["hm_sigmoid"] - ref_hm).max())
    sc = rdet[..., 4]
    gap_ok = np.ones_like(sc, bool)
    gap_ok[:, :-1] &= (sc[:, :-1] - sc[:, 1:]) > 4 * err
    gap_ok[:, 1:] &= (sc[:, :-1] - sc[:, 1:]) > 4 * err
    assert gap_ok.mean() > 0.5
    assert np.array_equal(inds[gap_ok], rinds[gap_ok])
    np.testing.assert_allclose(dets[gap_ok], rdet[gap_ok], atol=1e-3, rtol=1e-3)
    eng.close()


def test_batch64_tolerance_mode_vs_cpu_oracle():
    """The tolerance mode AT THE BATCH IT IS TIMED ON (BASELINE configs[1]: B = 64, 640x640, top-100): the
    fp32_split engine on the bench's 64 synthetic images against the CPU oracle -- the reference's fp32 arithmetic -- on four
    images spread over the batch (every image runs the same kernels, the batch index is only blockIdx.z; the oracle of all 64
    takes minutes).  north_star's tolerance, written here: head maps and sigmoid scores rtol = atol = 1e-3; decoded boxes and
    scores within 1e-3; top-100 indices identical wherever the oracle's neighbouring scores are separated by more than 4x the
    measured score error (gap-qualified ranks), and at the other ranks the engine's index is a near-tie of the oracle's."""
    B, S, K = 64, 640, 100
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    sd = cfa.weights.synthetic_state_dict(0)
    tsd = O.to_torch_sd(sd)
    eng = cfa.Engine(S, S, max_batch=B, dtype="fp32_split", weights=sd)
    eng.forward_enqueue(imgs)
    got = eng.heads(sigmoid_hm=True)
    dets, lms, inds = eng.decode_topk(K=K)
    pick = [0, 21, 42, 63]
    # decode of OUR heads is bit-exact against the oracle decode of the same heads
    od, ol, oi = O.ctdet_decode(got["hm_sigmoid"][pick], got["wh"][pick], got["reg"][pick], K, got["lm"][pick])
    assert np.array_equal(inds[pick], oi) and np.array_equal(dets[pick], od) and np.array_equal(lms[pick], ol)
    n_gap = n_same = 0
    for b in pick:
        ref = O.forward(tsd, torch.from_numpy(O.preprocess(imgs[b])))
        for k in ("hm", "wh", "lm", "reg"):
            np.testing.assert_allclose(got[k][b:b + 1], ref[k].numpy(), rtol=1e-3, atol=1e-3, err_msg="image %d head %s" % (b, k))
        ref_hm = O.sigmoid_clamp(ref["hm"]).numpy()
        np.testing.assert_allclose(got["hm_sigmoid"][b:b + 1], ref_hm, atol=1e-3, rtol=0)
        rdet, rlms, rinds = O.ctdet_decode(ref_hm, ref["wh"].numpy(), ref["reg"].numpy(), K, ref["lm"].numpy())
        err = float(np.abs(got["hm_sigmoid"][b:b + 1] - ref_hm).max())
        sc = rdet[0, :, 4]
        gap_ok = np.ones_like(sc, bool)
        gap_ok[:-1] &= (sc[:-1] - sc[1:]) > 4 * err
        gap_ok[1:] &= (sc[:-1] - sc[1:]) > 4 * err
        assert gap_ok.mean() > 0.5, (b, float(gap_ok.mean()), err)
        assert np.array_equal(inds[b][gap_ok], rinds[0][gap_ok]), b
        np.testing.assert_allclose(dets[b][gap_ok], rdet[0][gap_ok], atol=1e-3, rtol=1e-3)
        np.testing.assert_allclose(lms[b][gap_ok], rlms[0][gap_ok], atol=1e-3, rtol=1e-3)
        same = inds[b] == rinds[0]
        flat = ref_hm.ravel()
        for r in np.nonzero(~same)[0]:                       # a swapped rank is a near-tie of the oracle's own scores
            assert abs(flat[inds[b][r]] - sc[r]) <= 4 * err, (b, int(r), float(flat[inds[b][r]]), float(sc[r]), err)
        assert len(set(inds[b].tolist()) & set(rinds[0].tolist())) >= 99, b
        n_gap += int(gap_ok.sum()); n_same += int(same.sum())
    print("gap-qualified ranks %d / 400, identical ranks %d / 400" % (n_gap, n_same))
    eng.close()


@pytest.mark.parametrize("dtype", EXACT)
def test_vga_480x640_fp32_vs_oracle(dtype):
    """BASELINE config 4 geometry (VGA, non-square, multiples of 32): heads + D3 decode vs the oracle."""
    rng = np.random.default_rng(44)
    img = rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8)
    sd = cfa.weights.synthetic_state_dict(0)
    eng = cfa.Engine(480, 640, max_batch=2, dtype=dtype, weights=sd)
    eng.forward_enqueue(img)
    got = eng.heads(sigmoid_hm=True)
    ref = O.forward(O.to_torch_sd(sd), torch.from_numpy(np.concatenate([O.preprocess(im) for im in img])))
    for k in ("hm", "wh", "lm", "reg"):
        assert got[k].shape == tuple(ref[k].shape) == (2, {"hm": 1, "wh": 2, "lm": 10, "reg": 2}[k], 120, 160)
        np.testing.assert_allclose(got[k], ref[k].numpy(), rtol=1e-3, atol=1e-3, err_msg=k)
    dets, lms, inds = eng.decode_topk(K=100)
    odet, olms, oinds = O.ctdet_decode(got["hm_sigmoid"], got["wh"], got["reg"], 100, got["lm"])
    assert np.array_equal(inds, oinds) and np.array_equal(dets, odet) and np.array_equal(lms, olms)
    eng.close()


@pytest.mark.parametrize("dtype", EXACT)
def test_crowd_1280_topk1000(dtype):
    """BASELINE config 5 geometry: 1280x1280, heat map 320x320, K = 1000.  fp32 heads vs the oracle on one
    image; bf16 batch: decode of our heads bit-exact vs the oracle decode, sorted, deterministic."""
    rng = np.random.default_rng(45)
    sd = cfa.weights.synthetic_state_dict(0)
    img = rng.integers(0, 256, (1, 1280, 1280, 3), dtype=np.uint8)
    eng = cfa.Engine(1280, 1280, max_batch=1, dtype=dtype, weights=sd)
    eng.forward_enqueue(img)
    got = eng.heads(sigmoid_hm=True)
    ref = O.forward(O.to_torch_sd(sd), torch.from_numpy(O.preprocess(img[0])))
    for k in ("hm", "wh", "lm", "reg"):
        np.testing.assert_allclose(got[k], ref[k].numpy(), rtol=1e-3, atol=1e-3, err_msg=k)
    dets, lms, inds = eng.decode_topk(K=1000)
    odet, olms, oinds = O.ctdet_decode(got["hm_sigmoid"], got["wh"], got["reg"], 1000, got["lm"])
    assert np.array_equal(inds, oinds) and np.array_equal(dets, odet) and np.array_equal(lms, olms)
    eng.close()
    imgs = rng.integers(0, 256, (4, 1280, 1280, 3), dtype=np.uint8)
    eng = cfa.Engine(1280, 1280, max_batch=4, dtype="bf16", weights=sd)
    eng.forward_enqueue(imgs)
    d1, l1, i1 = eng.decode_topk(K=1000)
    hd = eng.heads(sigmoid_hm=True)
    od, ol, oi = O.ctdet_decode(hd["hm_sigmoid"], hd["wh"], hd["reg"], 1000, hd["lm"])
    assert np.array_equal(i1, oi) and np.array_equal(d1, od)
    assert (np.diff(d1[..., 4], axis=1) <= 0).all() and i1.max() < 320 * 320
    eng.close()


def test_device_output_decode_overlapped_stream():
    """cf_decode_topk with device destinations runs on the ctx's second stream (overlapping the next
    forward); results must equal the synchronous host-output path, run after run."""
    rng = np.random.default_rng(77)
    B, K = 4, 50
    eng = cfa.Engine(128, 160, max_batch=B, dtype="bf16")
    imgs = [rng.integers(0, 256, (B, 128, 160, 3), dtype=np.uint8) for _ in range(3)]
    d_dets = eng.device_alloc(B * K * 6 * 4); d_lms = eng.device_alloc(B * K * 10 * 4); d_inds = eng.device_alloc(B * K * 8)
    import ctypes
    outs = []
    for x in imgs + imgs:                      # back-to-back enqueues, no host sync in between
        eng.forward_enqueue(x)
        eng.decode_topk_device(K, d_dets, d_lms, d_inds)
    eng.synchronize()
    dets = np.empty((B, K, 6), np.float32); inds = np.empty((B, K), np.int64)
    eng._chk(eng._L.cf_memcpy_d2h(eng._h, cfa._lib.ptr(dets), ctypes.c_void_p(d_dets), dets.nbytes))
    eng._chk(eng._L.cf_memcpy_d2h(eng._h, cfa._lib.ptr(inds), ctypes.c_void_p(d_inds), inds.nbytes))
    eng.forward_enqueue(imgs[-1])
    hd, _, hi = eng.decode_topk(K)
    assert np.array_equal(inds, hi) and np.array_equal(dets, hd)
    for p_ in (d_dets, d_lms, d_inds):
        eng.device_free(p_)
    eng.close()


def test_hipgraph_replay_equals_eager_launches():
    """The backbone + neck launches are replayed from a hipGraph from the second forward with the same
    (input buffer, format, batch) on; results must be bit-identical to eager launches (CF_FLAG_NO_GRAPH),
    for a changing batch size and changing input contents, and the graph must really have been captured."""
    rng = np.random.default_rng(5)
    eg = cfa.Engine(96, 128, max_batch=4, dtype="bf16", graph=True)
    ee = cfa.Engine(96, 128, max_batch=4, dtype="bf16", graph=False)
    for it in range(6):
        B = (4, 2)[it % 2]
        x = rng.integers(0, 256, (B, 96, 128, 3), dtype=np.uint8)
        eg.forward_enqueue(x); ee.forward_enqueue(x)
        hg, he = eg.heads(), ee.heads()
        for k in ("hm", "wh", "lm", "reg"):
            assert np.array_equal(hg[k], he[k]), (it, k)
        dg, de = eg.decode_topk(20), ee.decode_topk(20)
        assert np.array_equal(dg[2], de[2]) and np.array_equal(dg[0], de[0])
    assert eg.graph_stats() == (2, 0), eg.graph_stats()      # B=4 and B=2 through the staging buffer
    assert ee.graph_stats() == (0, 0)
    eg.close(); ee.close()


def test_batch64_bf16_properties():
    """BASELINE config 2 at full size (B=64, 640x640, bf16): size-independent properties --
    batch-slot independence, run-to-run determinism, decode sortedness and index validity."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (4, 640, 640, 3), dtype=np.uint8)
    img = np.concatenate([base] * 16)                   # 64 images, period 4
    eng = cfa.Engine(640, 640, max_batch=64, dtype="bf16")
    eng.forward_enqueue(img)
    d1, l1, i1 = eng.decode_topk(K=100)
    eng.forward_enqueue(img)
    d2, l2, i2 = eng.decode_topk(K=100)
    assert np.array_equal(d1, d2) and np.array_equal(i1, i2) and np.array_equal(l1, l2)     # deterministic
    for r in range(1, 16):                                                                # slot independent
        assert np.array_equal(i1[:4], i1[4 * r:4 * r + 4]) and np.array_equal(d1[:4], d1[4 * r:4 * r + 4])
    assert (np.diff(d1[..., 4], axis=1) <= 0).all()
    assert i1.min() >= 0 and i1.max() < 160 * 160
    assert all(len(np.unique(row)) == 100 for row in i1[:4])
    # accuracy of this batch against the bf16-emulating oracle: tests/test_bf16_parity.py::test_batch64_end_to_end_vs_emulation
    eng.close()


@pytest.mark.parametrize("cfg", [(96, 5, 2, 40, 40), (96, 5, 2, 23, 31), (160, 5, 1, 20, 20), (160, 5, 1, 13, 27),
                                 (160, 3, 1, 20, 20), (160, 3, 1, 9, 41)])
def test_expand_dw_kernel_vs_oracle(cfg):
    """The expand+depthwise kernel of the wide late blocks (layer5.0 / 5.1 / 6.0 shapes; bf16 storage, fp16
    tile) against the oracle's fp32 expand -> Swish -> depthwise -> Swish, incl. odd map sizes (edge tiles,
    partly filled waves) and batch > 1."""
    cin, k, s, H, W = cfg
    rng = np.random.default_rng(cin + 10 * k + s + H)
    hid = cin * 6
    we = (rng.standard_normal((hid, cin, 1, 1)) * 1.5 / np.sqrt(cin)).astype(np.float32)
    wd = (rng.standard_normal((hid, 1, k, k)) * 1.5 / k).astype(np.float32)
    x = rng.standard_normal((2, cin, H, W)).astype(np.float32)
    t = torch.from_numpy(x)
    ref = O.conv_swish(O.conv_swish(t, torch.from_numpy(we), 1, 1), torch.from_numpy(wd), k, s, groups=hid).numpy()
    y = ops.expand_dw(x, we, wd, k, s, dtype="bf16")
    assert y.shape == ref.shape
    _emu_close(y, E.expand_dw(E.q_bf16(t), we.reshape(hid, cin), wd, k, s, out_scaled=False), cfg)
    d = np.abs(y - ref)                                # and the fp32 oracle at the bf16 noise floor
    assert d.max() < 0.08 and d.mean() < 0.006, (float(d.max()), float(d.mean()))
    with pytest.raises(ValueError):
        ops.expand_dw(x, we, wd, k, s, dtype="fp32")           # bf16-only kernel: loud, no fallback


@pytest.mark.parametrize("cfg", [(64, 5, 1, 40, 40), (96, 5, 1, 40, 40), (96, 5, 1, 17, 45), (96, 5, 2, 40, 40), (96, 5, 2, 23, 31),
                                 (160, 5, 1, 20, 20), (160, 5, 1, 13, 27), (160, 3, 1, 20, 20), (160, 3, 1, 9, 41)])
def test_expand_dw_f32_kernel_vs_oracle(cfg):
    """The round-5 expand+depthwise kernel of the tolerance mode (cf_mbconv5.hip: fp32 storage, split-bf16 expand products, register-
    window depthwise on an x-quad-cell tile, packed taps) on every production shape (layer4.0 ... 6.0) and on map sizes that are
    not a multiple of the tile or of a strip of four (edge tiles, partly filled strip groups, stride-2 even / odd halves), batch
    > 1: against the oracle's fp32 expand -> Swish -> depthwise -> Swish at the per-op tolerance of the mode (1e-4 of the output
    scale).  The exact-fp32 mode keeps its round-4 kernels: asking for it here is an error, not a fallback."""
    cin, k, s, H, W = cfg
    rng = np.random.default_rng(cin + 10 * k + s + H)
    hid = cin * 6
    we = (rng.standard_normal((hid, cin, 1, 1)) * 1.5 / np.sqrt(cin)).astype(np.float32)
    wd = (rng.standard_normal((hid, 1, k, k)) * 1.5 / k).astype(np.float32)
    x = rng.standard_normal((3, cin, H, W)).astype(np.float32)
    ref = O.conv_swish(O.conv_swish(torch.from_numpy(x), torch.from_numpy(we), 1, 1), torch.from_numpy(wd), k, s, groups=hid).numpy()
    y = ops.expand_dw(x, we, wd, k, s, dtype="fp32_split")
    assert y.shape == ref.shape
    scale = float(np.sqrt((ref ** 2).mean()))
    np.testing.assert_allclose(y, ref, rtol=1e-4, atol=1e-4 * max(1.0, scale), err_msg=str(cfg))
    with pytest.raises(ValueError):
        ops.expand_dw(x, we, wd, k, s, dtype="fp32")


# (bf16 engine on sizes that are not multiples of any tile / smaller than a tile: tests/test_bf16_parity.py,
#  layer by layer against the emulating oracle -- it replaced the bf16-vs-fp32-engine noise-floor tests that stood here)


@pytest.mark.parametrize("dtype", ["fp32", "fp32_split", "bf16"])
def test_variable_size_buckets_match_per_shape_detectors(dtype):
    """BASELINE configs[3] (VGA-class images of different shapes in one batch): CenterFaceBuckets groups by
    network shape and must return, per image and in input order, exactly what CenterFace(h, w)(img) returns --
    in the fp32 parity mode and in the benchmarked bf16 mode (an image's result does not depend on its batch:
    the batch index is only blockIdx.z)."""
    rng = np.random.default_rng(2024)
    shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416), (478, 720), (300, 500), (470, 730), (630, 470)]
    imgs = [rng.integers(0, 256, shapes[i % len(shapes)] + (3,), dtype=np.uint8) for i in range(21)]
    pool = cfa.CenterFaceBuckets(dtype=dtype, max_batch=4, max_buckets=4)        # fewer contexts than shapes: eviction
    got = pool.detect(imgs)
    assert len(got) == len(imgs)
    # raw sizes that round up to the same network shape share one context: (478,720)/(470,730) -> 480x736,
    # (640,480)/(630,470) -> 640x480; 9 raw shapes = 7 network shapes
    roomy = cfa.CenterFaceBuckets(dtype=dtype, max_batch=4, max_buckets=16)
    got2 = roomy.detect(imgs)
    assert roomy.created == 7
    for a, b2 in zip(got, got2):
        assert np.array_equal(a[0], b2[0]) and np.array_equal(a[1], b2[1])
    roomy.close()
    for (h, w) in shapes:
        one = cfa.CenterFace(h, w, dtype=dtype)
        for i, im in enumerate(imgs):
            if im.shape[:2] != (h, w):
                continue
            d, l = one(im)
            assert d.shape == got[i][0].shape and np.array_equal(d, got[i][0]) and np.array_equal(l, got[i][1]), (i, h, w)
        one.close()
    pool.close()


@pytest.mark.isolated
def test_variable_size_buckets_product_configuration_128_images():
    """The configs[3] PRODUCT configuration: 128 images of five VGA shapes through CenterFaceBuckets(bf16, max_batch=32)
    -- chunks of 32, page-locked staging reused across chunks and shapes (one grow-only buffer per context), resized
    and native-size chunks mixed -- against the one-image-per-call detector of each shape on a sample of the batch."""
    rng = np.random.default_rng(7)
    shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416)]
    order = rng.integers(0, len(shapes), 128)
    imgs = [rng.integers(0, 256, shapes[k] + (3,), dtype=np.uint8) for k in order]
    pool = cfa.CenterFaceBuckets(dtype="bf16", max_batch=32, max_buckets=4)
    got = pool.detect(imgs)
    got_again = pool.detect(imgs)                       # contexts, graphs and staging buffers reused
    assert len(got) == 128
    for a, b2 in zip(got, got_again):
        assert np.array_equal(a[0], b2[0]) and np.array_equal(a[1], b2[1])
    for si, (h, w) in enumerate(shapes):
        one = cfa.CenterFace(h, w, dtype="bf16")
        idx = [i for i in range(128) if order[i] == si][::5]
        assert idx
        for i in idx:
            d, l = one(imgs[i])
            assert d.shape == got[i][0].shape and np.array_equal(d, got[i][0]) and np.array_equal(l, got[i][1]), (i, h, w)
        one.close()
    pool.close()


def test_device_rescale_equals_numpy_floor_divide():
    """centerface.py:55-62 inside the decode kernel (cf_set_rescale): bit-identical to numpy's float32 `//` by the python-float
    scales of transform(), applied to the same engine's un-rescaled decode -- on scales that are not representable (480/470,
    736/730), on scale 1.0 (still a floor) and after switching it off again."""
    rng = np.random.default_rng(99)
    total = 0
    for (h, w) in [(470, 730), (640, 640), (300, 500)]:
        H, W, sh, sw = cfa.CenterFace.transform(None, h, w)
        eng = cfa.Engine(H, W, max_batch=3, dtype="fp32")
        x = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
        eng.forward_resized_enqueue(x)
        plain = eng.decode_threshold(0.3, 0.3, 1024)
        eng.set_rescale(sh, sw)
        eng.decode_threshold_enqueue(0.3, 0.3, 1024)            # the pre-enqueued launch carries the scales too
        scaled = eng.decode_threshold(0.3, 0.3, 1024)
        eng.set_rescale(0.0, 0.0)
        off = eng.decode_threshold(0.3, 0.3, 1024)
        for (d, l), (ds, ls), (d0, l0) in zip(plain, scaled, off):
            want_d, want_l = d.copy(), l.copy()
            want_d[:, 0:4:2], want_d[:, 1:4:2] = d[:, 0:4:2] // sw, d[:, 1:4:2] // sh            # the reference's own statement
            want_l[:, 0:10:2], want_l[:, 1:10:2] = l[:, 0:10:2] // sw, l[:, 1:10:2] // sh
            assert np.array_equal(ds, want_d) and np.array_equal(ls, want_l)
            assert np.array_equal(d0, d) and np.array_equal(l0, l)
            total += len(d)
        with pytest.raises(ValueError):
            eng.set_rescale(1.0, 0.0)
        eng.close()
    assert total > 0


def _mapped_pinned_copy(im):
    """A frame-pool style buffer: an anonymous mmap region of whole pages (a mapping of its own -- the only kind of caller memory
    cfa.pin accepts), page-locked in place; returns (image view, the registered base array)."""
    import mmap
    n = im.nbytes
    base = np.frombuffer(mmap.mmap(-1, (n + 4095) // 4096 * 4096), np.uint8)
    cfa.pin(base)
    view = base[:n].reshape(im.shape)
    view[...] = im
    return view, base


@pytest.mark.isolated
def test_pin_refuses_heap_memory_and_partial_pages():
    """Round 6 (the abort of GPUTEST_r05): numpy heap arrays are never page-locked in place.  cfa.pin accepts whole pages of a mapping of
    its own and refuses everything else with a ValueError; cf_host_register (C ABI) refuses unaligned ranges; pinned_empty /
    pinned_copy hand out hipHostMalloc memory that is released with its last view."""
    import ctypes as C
    import gc
    import mmap
    L = cfa._lib.lib()
    small = np.zeros(9 * 4096, np.uint8)                             # 36 KB: from the brk heap; a page-aligned window of whole pages in it
    small = small[(-small.ctypes.data) % 4096:][:8 * 4096]
    for arr in (np.zeros((480, 640, 3), np.uint8), np.zeros(4096 * 4 + 16, np.uint8)[16:], np.zeros(100, np.uint8), small):
        with pytest.raises(ValueError, match="pinned_empty"):
            cfa.pin(arr)
        assert not cfa.is_pinned(arr)
    with pytest.raises(ValueError):
        cfa.pin(np.zeros((4, 4), np.uint8)[:, ::2])                 # not C-contiguous
    buf = np.zeros(3 * 4096, np.uint8)
    al = (-buf.ctypes.data) % 4096
    assert L.cf_host_register(C.c_void_p(buf.ctypes.data + al + 8), 4096) == -1 and b"whole pages" in L.cf_op_last_error()
    assert L.cf_host_register(C.c_void_p(buf.ctypes.data + al), 4000) == -1
    assert L.cf_host_register(None, 4096) == -1 and L.cf_host_unregister(None) == -1
    assert L.cf_host_register(C.c_void_p(buf.ctypes.data + al), 0) == -1
    base = np.frombuffer(mmap.mmap(-1, 8 * 4096), np.uint8)
    assert cfa.pin(base) is base and cfa.is_pinned(base[100:200]) and cfa.pin(base) is base      # idempotent
    with pytest.raises(ValueError, match="overlaps"):
        cfa.pin(base[4096:8192])
    cfa.unpin(base)
    assert not cfa.is_pinned(base)
    a = cfa.pinned_empty((37, 53, 3))
    b = cfa.pinned_copy(np.arange(24, dtype=np.float32).reshape(2, 3, 4))
    assert a.shape == (37, 53, 3) and a.dtype == np.uint8 and cfa.is_pinned(a) and cfa.is_pinned(a[3])
    assert b.dtype == np.float32 and np.array_equal(b, np.arange(24, dtype=np.float32).reshape(2, 3, 4)) and cfa.is_pinned(b)
    addr = a.ctypes.data
    row = a[5]
    del a
    gc.collect()
    assert cfa.is_pinned(row)                                       # a view keeps the block alive
    del row
    gc.collect()
    with cfa.centerface._pin_lock:
        assert addr not in cfa.centerface._pin_sizes                # released with the last view
    p = C.c_void_p()
    assert L.cf_pinned_alloc(0, C.byref(p)) == -1 and L.cf_pinned_free(None) == 0


@pytest.mark.isolated
def test_pinned_images_reach_the_gpu_without_staging_and_match():
    """Page-locked caller images (cfa.pinned_empty, or cfa.pin on a mapped frame pool) go through cf_forward_images -- one DMA per image, no
    staging copy -- in CenterFaceBuckets and CenterFace.detect_batch, with the results of the staged path; pageable images keep the staged path."""
    rng = np.random.default_rng(5)
    shapes = [(480, 640), (640, 480), (640, 640), (470, 730)]
    pageable = [rng.integers(0, 256, shapes[i % 4] + (3,), dtype=np.uint8) for i in range(22)]
    pinned, bases = [], []
    for k, im in enumerate(pageable):
        if k % 2:
            a = cfa.pinned_empty(im.shape)
            a[...] = im
        else:
            a, base = _mapped_pinned_copy(im)                        # a caller's own mapping, page-locked in place
            bases.append(base)
        assert cfa.is_pinned(a) and cfa.is_pinned(a[10:20]) and not cfa.is_pinned(im)
        pinned.append(a)
    calls = {"direct": 0, "staged": 0}
    real_i, real_f, real_r = cfa.Engine.forward_images_enqueue, cfa.Engine.forward_enqueue, cfa.Engine.forward_resized_enqueue

    def count(key, f):
        def g(self, *a, **k):
            calls[key] += 1
            return f(self, *a, **k)
        return g
    cfa.Engine.forward_images_enqueue = count("direct", real_i)
    real_u = cfa.Engine._upload_addrs
    cfa.Engine._upload_addrs = count("direct", real_u)                 # CenterFaceBuckets' validated fast path into cf_upload_images
    cfa.Engine.forward_enqueue, cfa.Engine.forward_resized_enqueue = count("staged", real_f), count("staged", real_r)
    try:
        with cfa.CenterFaceBuckets(dtype="bf16", max_batch=4, max_buckets=4) as pool:
            want = pool.detect(pageable)
            assert calls["direct"] == 0 and calls["staged"] > 0
            calls["staged"] = 0
            got = pool.detect(pinned)
            assert calls["staged"] == 0 and calls["direct"] > 0
            mixed = pool.detect([pinned[i] if i % 3 else pageable[i] for i in range(22)])       # a chunk with one pageable image is staged
        one = cfa.CenterFace(470, 730, dtype="bf16", max_batch=4)
        sel = [i for i in range(22) if pageable[i].shape[:2] == (470, 730)]
        calls["direct"] = calls["staged"] = 0
        a = one.detect_batch([pageable[i] for i in sel])
        b = one.detect_batch([pinned[i] for i in sel])
        assert calls["direct"] > 0 and calls["staged"] > 0
        one.close()
    finally:
        cfa.Engine.forward_images_enqueue, cfa.Engine.forward_enqueue, cfa.Engine.forward_resized_enqueue = real_i, real_f, real_r
        cfa.Engine._upload_addrs = real_u
    assert sum(len(r[0]) for r in want) > 0
    for w_, g_, m_ in zip(want, got, mixed):
        assert np.array_equal(w_[0], g_[0]) and np.array_equal(w_[1], g_[1])
        assert np.array_equal(w_[0], m_[0]) and np.array_equal(w_[1], m_[1])
    for (d1, l1), (d2, l2), i in zip(a, b, sel):
        assert np.array_equal(d1, d2) and np.array_equal(l1, l2)
        assert np.array_equal(d1, want[i][0]) and np.array_equal(l1, want[i][1])
    for base in bases:
        cfa.unpin(base)
        assert not cfa.is_pinned(base)


@pytest.mark.isolated
def test_upload_forward_split_and_its_error_paths():
    """cf_upload_images / cf_forward_uploaded / cf_forward_images through the C ABI: the split call equals the block call bit for bit
    (network-sized and resized, page-locked and pageable pointers, images adjacent in memory = one DMA run, a batch large enough for
    the shared copy streams); a forward without an upload is CF_ESTATE, an upload replaced by another forward is gone, null images and
    oversized batches are CF_EINVAL."""
    import ctypes as C
    L = cfa._lib.lib()
    rng = np.random.default_rng(17)
    for (h, w, H, W, B) in [(64, 96, 64, 96, 3), (50, 70, 64, 96, 3), (640, 640, 640, 640, 9)]:
        eng = cfa.Engine(H, W, max_batch=B, dtype="bf16")
        block = cfa.pinned_empty((B, h, w, 3))                      # adjacent images: cf_upload_images coalesces them into one copy
        block[...] = rng.integers(0, 256, block.shape, dtype=np.uint8)
        loose = [block[b].copy() for b in range(B)]                 # pageable, separately allocated
        mapped = [_mapped_pinned_copy(block[b]) for b in range(B)]   # page-locked in place, separately mapped (a caller's frame pool)
        pinned = [m[0] for m in mapped]
        if (h, w) == (H, W):
            eng.forward_enqueue(block)
        else:
            eng.forward_resized_enqueue(block)
        want = eng.heads()
        for imgs in ([block[b] for b in range(B)], loose, pinned):
            eng.forward_images_enqueue(imgs)
            got = eng.heads()
            for k in want:
                assert np.array_equal(want[k], got[k]), (k, h, w)
            eng.upload_images(imgs)
            eng.forward_uploaded()
            got = eng.heads()
            for k in want:
                assert np.array_equal(want[k], got[k]), (k, h, w)
        with pytest.raises(cfa._lib.CenterFaceError, match="without a cf_upload_images"):
            eng.forward_uploaded()
        eng.upload_images(pinned)
        eng.forward_images_enqueue(loose)                           # another forward: the pending upload is dropped, not run later
        with pytest.raises(cfa._lib.CenterFaceError, match="without a cf_upload_images"):
            eng.forward_uploaded()
        ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in pinned])
        assert L.cf_upload_images(eng._h, ptrs, B + 1, h, w) == -1
        ptrs[B - 1] = None
        assert L.cf_upload_images(eng._h, ptrs, B, h, w) == -1 and b"null pointer" in L.cf_last_error(eng._h)
        assert L.cf_forward_uploaded(None) == -1 and L.cf_upload_images(None, ptrs, B, h, w) == -1
        eng.synchronize()
        for m in mapped:
            cfa.unpin(m[1])
        eng.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp32_split"])
@pytest.mark.parametrize("size,B", [((96, 128), 3), ((160, 224), 2), ((352, 640), 2), ((32, 32), 1), ((64, 416), 2)])
def test_fused_neck_bit_equal_to_three_kernels(size, B, dtype):
    """cf_neck.hip (conv_last + up1 + up2 as one launch, the 1/32 and 1/16 maps only in LDS) performs the same arithmetic in the
    same order as three pw_kernel launches (bf16, and the split-bf16 tolerance mode with fp32 tiles): the up2 tensor and everything after it are bit-identical (maps that are not multiples
    of the 2x4-cell tile, a one-cell map, batches)."""
    H, W = size
    rng = np.random.default_rng(H * 7 + W)
    x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    ef = cfa.Engine(H, W, max_batch=B, dtype=dtype, neck=True)
    e3 = cfa.Engine(H, W, max_batch=B, dtype=dtype, neck=False)
    pf, p3 = ef.plan(), e3.plan()
    assert [o["name"] for o in pf if not o["fused_away"]].count("conv_last+up1+up2") == 1
    assert any(o["name"] == "conv_last" and not o["fused_away"] for o in p3)
    i_f = [o["index"] for o in pf if o["name"] == "conv_last+up1+up2"][0]
    i_3 = [o["index"] for o in p3 if o["name"] == "up2"][0]
    assert np.array_equal(ef.trace(x, i_f), e3.trace(x, i_3))
    ef.forward_enqueue(x); e3.forward_enqueue(x)
    hf, h3 = ef.heads(), e3.heads()
    for k in hf:
        assert np.array_equal(hf[k], h3[k]), k
    for a, b2 in zip(ef.decode_topk(20), e3.decode_topk(20)):
        assert np.array_equal(a, b2)
    ef.close(); e3.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp32_split"])
@pytest.mark.parametrize("size", [(96, 128), (160, 224), (352, 640)])
def test_fused_up3_heads_bit_equal_to_two_kernels(size, dtype):
    """cf_uphead.hip (last IDAUp stage + collapsed heads, neck output only in LDS) performs the same arithmetic
    in the same order as cf_pw.hip's IDAUp epilogue followed by cf_head.hip: bit-identical head maps, on map
    sizes with partial tiles in both directions."""
    H, W = size
    rng = np.random.default_rng(H + W)
    x = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
    ef = cfa.Engine(H, W, max_batch=3, dtype=dtype, uphead=True)
    e2 = cfa.Engine(H, W, max_batch=3, dtype=dtype, uphead=False)
    assert any(op["name"] == "up3+heads" for op in ef.plan()) and not any(op["name"] == "up3+heads" for op in e2.plan())
    ef.forward_enqueue(x); e2.forward_enqueue(x)
    hf, h2 = ef.heads(sigmoid_hm=True), e2.heads(sigmoid_hm=True)
    for k in ("hm", "wh", "lm", "reg", "hm_sigmoid"):
        assert np.array_equal(hf[k], h2[k]), k
    df, d2 = ef.decode_topk(50), e2.decode_topk(50)
    assert np.array_equal(df[2], d2[2]) and np.array_equal(df[0], d2[0])
    ef.close(); e2.close()


# ----------------------------------------------------------------------------- N4: training-side pieces
def _train_batch(g, idx):
    st = lambda key: np.stack([g["b%d_%s" % (i, key)] for i in idx])
    return {"hm": st("hm"), "reg_mask": st("reg_mask"), "ind": st("ind"), "wh": st("wh"), "reg": st("reg"),
            "lm_mask": st("lm_mask"), "lm_ind": st("lm_ind"), "lm": st("landmarks")}


def test_target_encoder_kernel_vs_reference_goldens(golden):
    """cf_op_encode_targets (cf_loss.hip) against the reference's gaussian_radius / draw_umich_gaussian inside the
    dataset loop (tests/golden/train.npz): integer outputs and fp32 targets bit-exact; the Gaussian heat map to
    1 float32 ulp (float64 exp on the device vs glibc)."""
    from centerface_amd import losses
    g = golden("train")
    H, W = g["b0_hm"].shape[1:]
    boxes = np.stack([g["b%d_boxes" % b] for b in range(3)])
    lms = np.stack([g["b%d_lms" % b] for b in range(3)])
    counts = np.array([int(g["b%d_n" % b]) for b in range(3)], np.int32)
    t = losses.encode_targets(boxes, lms, counts, H, W)
    for b in range(3):
        for k in ("wh", "reg", "ind", "reg_mask", "landmarks", "lm_ind", "lm_mask"):
            assert np.array_equal(t[k][b], g["b%d_%s" % (b, k)]), (b, k)
        np.testing.assert_allclose(t["hm"][b], g["b%d_hm" % b], rtol=2e-7, atol=1e-9)
        assert np.array_equal(t["hm"][b] == 1.0, g["b%d_hm" % b] == 1.0)           # peaks (the focal loss's positives)
    # a larger random case against the oracle
    rng = np.random.default_rng(9)
    B, M, h, w = 4, 32, 160, 160
    bx = np.zeros((B, M, 4), np.float32); lm = -np.ones((B, M, 10), np.float32)
    cnt = np.array([32, 17, 0, 5], np.int32)
    for b in range(B):
        for k in range(cnt[b]):
            cx, cy, bw, bh = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(0.5, 60), rng.uniform(0.5, 60)
            bx[b, k] = [cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2]
            if k % 2 == 0:
                lm[b, k, 0::2] = rng.uniform(bx[b, k, 0], bx[b, k, 2], 5); lm[b, k, 1::2] = rng.uniform(bx[b, k, 1], bx[b, k, 3], 5)
    t = losses.encode_targets(bx, lm, cnt, h, w)
    for b in range(B):
        r = O.encode_targets(bx[b, :cnt[b]], lm[b, :cnt[b]], h, w, M)
        for k in ("wh", "reg", "ind", "reg_mask", "landmarks", "lm_ind", "lm_mask"):
            assert np.array_equal(t[k][b], r[k]), (b, k)
        np.testing.assert_allclose(t["hm"][b], r["hm"], rtol=2e-7, atol=1e-9)


def test_target_encoder_kernel_vs_reference_getitem(golden):
    """losses.to_output_map + cf_op_encode_targets against what the reference's own ``CenterFaceData.__getitem__`` returned
    (tests/golden/train_getitem.npz, tools/gen_goldens_getitem.py): integer outputs and fp32 targets bit-exact, the Gaussian
    heat map to 1 float32 ulp with identical peaks."""
    from centerface_amd import losses
    from test_oracle_vs_golden import _getitem_inputs
    g = golden("train_getitem")
    n = int(g["n_samples"])
    bx = np.zeros((n, 128, 4), np.float32); lm = -np.ones((n, 128, 10), np.float32); cnt = np.zeros(n, np.int32)
    for i in range(n):
        w = int(g["s%d_size" % i][1])
        boxes, lms = _getitem_inputs(g, i)
        ob, ol = losses.to_output_map(boxes[:128], lms[:128], g["s%d_c" % i], float(g["s%d_s" % i]), 160, 160,
                                      flipped=bool(g["s%d_flipped" % i]), width=w)
        cnt[i] = len(ob); bx[i, :len(ob)] = ob; lm[i, :len(ob)] = ol
    t = losses.encode_targets(bx, lm, cnt, 160, 160)
    for i in range(n):
        for k, gk in (("wh", "wh"), ("reg", "reg"), ("ind", "ind"), ("reg_mask", "reg_mask"), ("landmarks", "lm"),
                      ("lm_ind", "lm_ind"), ("lm_mask", "lm_mask")):
            assert np.array_equal(t[k][i], g["s%d_%s" % (i, gk)]), (i, k)
        np.testing.assert_allclose(t["hm"][i], g["s%d_hm" % i], rtol=2e-7, atol=1e-9)
        assert np.array_equal(t["hm"][i] == 1.0, g["s%d_hm" % i] == 1.0)


def test_dataset_affine_and_rotated_transform_vs_oracle():
    """dataset/dataset.py:146,160-179 (boxes / landmarks -> output-map coordinates, incl. the flip) in front of the
    target encoder, and get_affine_transform with rotation and shift (utils/image.py:27-60), against the oracle's
    restatement (cv2.getAffineTransform is replaced by a float64 solve on both sides; pinned analytically in
    test_oracle_vs_golden.py::test_post_process_affine_analytic)."""
    from centerface_amd import losses, post_process as pp
    rng = np.random.default_rng(9)
    boxes = np.sort(rng.uniform(0, 600, (7, 2, 2)), axis=1).transpose(0, 1, 2).reshape(7, 4).astype(np.float32)
    boxes = boxes[:, [0, 1, 2, 3]]
    lms = rng.uniform(0, 600, (7, 10)).astype(np.float32)
    lms[2, 0] = -1.0                                                    # no landmarks on this face
    c, s = np.array([311.5, 287.0], np.float32), np.float32(731.0)
    for flipped in (False, True):
        gb, gl = losses.to_output_map(boxes, lms, c, s, 160, 160, flipped=flipped, width=640)
        rb, rl = O.dataset_to_output_map(boxes, lms, c, s, 160, 160, flipped=flipped, width=640)
        assert np.array_equal(gb, rb) and np.array_equal(gl, rl), flipped
    enc = losses.encode_targets(gb[None], gl[None], np.array([7], np.int32), 160, 160)
    ref = O.encode_targets(rb, rl, 160, 160, 7)
    assert np.array_equal(enc["ind"][0], ref["ind"]) and np.array_equal(enc["hm"][0], ref["hm"])
    for rot, shift, inv in ((0, None, 0), (30, None, 1), (-75.5, np.array([0.1, -0.2], np.float32), 0), (180, None, 1)):
        kw = {} if shift is None else {"shift": shift}
        got = pp.get_affine_transform(c, s, rot, [160, 120], inv=inv, **kw)
        want = O.get_affine_transform(c, s, rot, [160, 120], inv=inv, **kw)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    # rot = 0 inverse: the library's matrix (the one the decode kernel applies) equals the general host solve
    np.testing.assert_allclose(pp.get_affine_transform(c, s, 0, [160, 120], inv=1), O.get_affine_transform(c, s, 0, [160, 120], inv=1),
                               rtol=0, atol=1e-9)


def test_ctdet_loss_kernel_vs_reference_goldens(golden):
    """cf_op_ctdet_loss against model/losses.py CtdetLoss outputs (focal + 3 x RegL1), incl. the num_pos == 0 branch;
    fp32 terms, double sums: 2e-5 relative."""
    from centerface_amd import losses
    g = golden("train")
    heads = {k: g["heads_" + k] for k in ("hm", "wh", "reg", "lm")}
    for name in ("all", "empty"):
        idx = [int(i) for i in g["loss_%s_idx" % name]]
        got = losses.ctdet_loss({k: v[idx] for k, v in heads.items()}, _train_batch(g, idx))
        np.testing.assert_allclose(got, g["loss_" + name], rtol=2e-5, atol=1e-6)
    got = losses.ctdet_loss({k: v[[0, 1]] for k, v in heads.items()}, _train_batch(g, [0, 1]), hm_w=0.5, wh_w=1.0, off_w=2.0, lm_w=0.25)
    ref = O.ctdet_loss({k: torch.from_numpy(v[[0, 1]].copy()) for k, v in heads.items()},
                       {k: torch.from_numpy(v) for k, v in _train_batch(g, [0, 1]).items()}, 0.5, 1.0, 2.0, 0.25)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6)


def test_ctdet_loss_on_engine_heads_matches_explicit_maps():
    """cf_ctdet_loss evaluates the loss on the head maps that stay on the GPU after a forward: same numbers as
    the op-level entry fed with cf_get_heads' copies, and as the oracle."""
    from centerface_amd import losses
    rng = np.random.default_rng(31)
    H, W, B, M = 128, 160, 3, 16
    eng = cfa.Engine(H, W, max_batch=B, dtype="fp32")
    eng.forward_enqueue(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8))
    hd = eng.heads()
    h, w = H // 4, W // 4
    bx = np.zeros((B, M, 4), np.float32); lm = -np.ones((B, M, 10), np.float32); cnt = np.array([9, 16, 3], np.int32)
    for b in range(B):
        for k in range(cnt[b]):
            cx, cy, bw, bh = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(1, 14), rng.uniform(1, 14)
            bx[b, k] = [cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2]
            lm[b, k, 0::2] = rng.uniform(bx[b, k, 0], bx[b, k, 2], 5); lm[b, k, 1::2] = rng.uniform(bx[b, k, 1], bx[b, k, 3], 5)
    t = losses.encode_targets(bx, lm, cnt, h, w)
    batch = {"hm": t["hm"], "reg_mask": t["reg_mask"], "ind": t["ind"], "wh": t["wh"], "reg": t["reg"],
             "lm_mask": t["lm_mask"], "lm_ind": t["lm_ind"], "lm": t["landmarks"]}
    on_gpu = losses.ctdet_loss_last_forward(eng, batch)
    explicit = losses.ctdet_loss(hd, batch)
    assert np.array_equal(on_gpu, explicit)
    ref = O.ctdet_loss({k: torch.from_numpy(hd[k].copy()) for k in ("hm", "wh", "reg", "lm")}, {k: torch.from_numpy(v) for k, v in batch.items()})
    np.testing.assert_allclose(on_gpu, ref, rtol=2e-5, atol=1e-6)
    eng.close()


@pytest.mark.isolated
def test_graph_cache_eviction_and_context_churn():
    """More (batch size) keys than the 16-entry hipGraph cache holds: evicted graphs are rebuilt, results stay
    bit-identical to eager launches; then 12 create/destroy cycles of contexts of different shapes (no leaked
    streams / graphs / buffers that would make a later cf_create fail)."""
    rng = np.random.default_rng(12)
    H, W, MB = 64, 96, 20
    eg = cfa.Engine(H, W, max_batch=MB, dtype="bf16", graph=True)
    ee = cfa.Engine(H, W, max_batch=MB, dtype="bf16", graph=False)
    x = rng.integers(0, 256, (MB, H, W, 3), dtype=np.uint8)
    dg, de = eg.device_alloc(x.nbytes), ee.device_alloc(x.nbytes)
    eg.memcpy_h2d(dg, x); ee.memcpy_h2d(de, x)
    for rep in range(2):
        for B in range(1, MB + 1):
            for again in range(3):                       # second sighting captures, third replays; 20 keys > 16 slots
                eg.forward_enqueue(dg, on_device=True, B=B, in_format=0)
                ee.forward_enqueue(de, on_device=True, B=B, in_format=0)
            if B % 5 == 0:
                assert np.array_equal(eg.heads()["hm"], ee.heads()["hm"]), B
    ng, nb = eg.graph_stats()
    assert 1 <= ng <= 16 and nb == 0, (ng, nb)
    eg.close(); ee.close()
    for i in range(12):
        e = cfa.Engine(32 * (1 + i % 4), 32 * (2 + i % 3), max_batch=2, dtype=("bf16", "fp32")[i % 2])
        e.forward_enqueue(rng.integers(0, 256, (2, e.H, e.W, 3), dtype=np.uint8))
        assert np.isfinite(e.heads()["hm"]).all()
        e.close()


def test_bf16_forward_is_bit_reproducible():
    """Run-to-run determinism of the throughput path (no atomics in the network, fixed reduction orders, exact
    top-K): ten forwards of the same resident batch give bit-identical head maps and detections."""
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (8, 320, 448, 3), dtype=np.uint8)
    e = cfa.Engine(320, 448, max_batch=8, dtype="bf16")
    d = e.device_alloc(x.nbytes); e.memcpy_h2d(d, x)
    ref = None
    for it in range(10):
        e.forward_enqueue(d, on_device=True, B=8, in_format=0)
        dets, lms, inds = e.decode_topk(64)
        cur = (dets.tobytes(), lms.tobytes(), inds.tobytes(), e.heads()["lm"].tobytes())
        ref = ref or cur
        assert cur == ref, it
    e.close()


# ----------------------------------------------------------------------------- round-2 hardening
@pytest.mark.isolated
def test_threshold_decode_takes_any_candidate_count_and_reports_truncation():
    """The reference's decode handles any number of cells above the threshold (centerface.py:78-79).  An early-training
    heat map (hm bias -1.79 -> sigmoid 0.143) with a low threshold puts most of a 160x160 map above it: the candidate
    workspace starts at 4096 per image and must grow (one retry) instead of failing with CF_EOVERFLOW; more survivors
    than `max_out` rows are reported through counts (the Python host then retries with enough rows)."""
    rng = np.random.default_rng(77)
    B, h, w = 2, 160, 160
    hm = np.clip(0.143 + 0.05 * rng.standard_normal((B, 1, h, w)), 1e-4, 1 - 1e-4).astype(np.float32)   # ~80 % of 25600 cells > 0.1
    wh = rng.uniform(0.5, 3.0, (B, 2, h, w)).astype(np.float32)                                          # small boxes: little suppression
    reg = rng.uniform(0, 1, (B, 2, h, w)).astype(np.float32)
    from centerface_amd import eval_widerface as ew
    for b in range(B):
        got = ew.decode(hm[b], wh[b], reg[b], None, (640, 640), threshold=0.1, nms_thresh=0.3)
        ref = O.decode_d2(hm[b], wh[b], reg[b], (640, 640), threshold=0.1)
        assert len(ref) > 4096, len(ref)                                     # beyond the initial capacity AND the initial max_out
        assert np.array_equal(np.asarray(got, np.float32), np.asarray(ref, np.float32))
    # raw C ABI: truncation is visible in counts
    import ctypes as C
    L = cfa._lib.lib()
    dets = np.empty((1, 100, 5), np.float32); cnt = np.zeros(1, np.int32)
    rc = L.cf_op_decode_threshold_ex(0, 1, cfa._lib.ptr(hm[:1]), cfa._lib.ptr(wh[:1]), cfa._lib.ptr(reg[:1]), None, 1, h, w, 640, 640,
                                     C.c_float(0.1), C.c_float(0.3), 100, cfa._lib.ptr(dets), None, cfa._lib.ptr(cnt))
    assert rc == 0 and cnt[0] == len(O.decode_d2(hm[0], wh[0], reg[0], (640, 640), threshold=0.1)) > 100
    assert np.array_equal(dets[0], np.asarray(O.decode_d2(hm[0], wh[0], reg[0], (640, 640), threshold=0.1), np.float32)[:100])


@pytest.mark.isolated
def test_threshold_decode_enqueue_then_collect():
    """cf_decode_threshold_enqueue: the decode kernels go into the stream right behind the forward; the collecting call with the
    same parameters returns exactly what a plain decode returns, other parameters or a newer forward make it launch its own."""
    S, B = 160, 5
    rng = np.random.default_rng(3)
    x1 = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    x2 = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    eng = cfa.Engine(S, S, max_batch=B, dtype="bf16")

    def same(a, b):
        return len(a) == len(b) and all(np.array_equal(p[0], q[0]) and np.array_equal(p[1], q[1]) for p, q in zip(a, b))
    eng.forward_enqueue(x1); want1 = eng.decode_threshold(0.3, 0.3, 256)
    want1_d2 = eng.decode_threshold(0.25, 0.3, 256, mode="d2", size=(640, 640))
    eng.forward_enqueue(x2); want2 = eng.decode_threshold(0.3, 0.3, 256)
    assert sum(len(d) for d, _ in want1) > 0 and not same(want1, want2)
    eng.forward_enqueue(x1); eng.decode_threshold_enqueue(0.3, 0.3, 256)
    assert same(eng.decode_threshold(0.3, 0.3, 256), want1)                      # collected
    assert same(eng.decode_threshold(0.3, 0.3, 256), want1)                      # and again: a fresh launch, same result
    eng.forward_enqueue(x1); eng.decode_threshold_enqueue(0.3, 0.3, 256)
    assert same(eng.decode_threshold(0.25, 0.3, 256, mode="d2", size=(640, 640)), want1_d2)      # other parameters: its own decode
    eng.forward_enqueue(x1); eng.decode_threshold_enqueue(0.3, 0.3, 256)
    eng.forward_enqueue(x2)                                                      # a newer forward invalidates the enqueued decode
    assert same(eng.decode_threshold(0.3, 0.3, 256), want2)
    eng.forward_enqueue(x1); eng.decode_threshold_enqueue(0.3, 0.3, 4)           # more survivors than rows: the grow-and-rerun path
    assert same(eng.decode_threshold(0.3, 0.3, 4), want1)
    eng.close()


@pytest.mark.isolated
def test_stream_and_buffer_hazards_between_entry_points():
    """(1) forward_resized followed by a host-input forward WITHOUT a sync in between: the resize writes a dedicated
    buffer, so the second call's H2D copy cannot overwrite what the first forward's stem still reads.  (2) A
    device-output top-K decode (decode stream) followed by a host-output decode (main stream): they share the key
    list, the second must wait for the first."""
    import torch
    rng = np.random.default_rng(3)
    H, W, B = 160, 224, 4
    small = rng.integers(0, 256, (B, 97, 131, 3), dtype=np.uint8)
    full = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    eng = cfa.Engine(H, W, max_batch=B, dtype="bf16")
    ref_eng = cfa.Engine(H, W, max_batch=B, dtype="bf16")
    ref_eng.forward_resized_enqueue(small); want_small = ref_eng.heads()
    ref_eng.forward_enqueue(full); want_full = ref_eng.heads()
    for _ in range(3):
        eng.forward_resized_enqueue(small)
        d_small = torch.empty((B, 20, 6), dtype=torch.float32, device="cuda")
        eng.decode_topk_device(20, d_small.data_ptr())                          # decode stream, asynchronous
        eng.forward_enqueue(full)                                               # no sync: H2D on the copy stream
        got_full = eng.heads()
        for k in ("hm", "wh", "lm", "reg"):
            assert np.array_equal(got_full[k], want_full[k]), k
        eng.synchronize()
        ref_eng.forward_resized_enqueue(small)
        assert np.array_equal(d_small.cpu().numpy(), ref_eng.decode_topk(20)[0])
    # (2) same forward decoded twice, back to back, through both streams
    eng.forward_enqueue(full)
    d_dev = torch.empty((B, 50, 6), dtype=torch.float32, device="cuda")
    for _ in range(4):
        eng.decode_topk_device(50, d_dev.data_ptr())
        d_host = eng.decode_topk(50)[0]
        eng.synchronize()
        assert np.array_equal(d_dev.cpu().numpy(), d_host)
    eng.close(); ref_eng.close()


@pytest.mark.isolated
def test_two_contexts_driven_from_two_threads():
    """include/centerface_hip.h: "different ctxs may be driven from different threads".  Two threads, each with its own
    Engine (different shapes and modes), run forwards, both decoders and a weight reload concurrently; every result must
    equal the single-threaded result of the same calls."""
    import threading
    rng = np.random.default_rng(21)
    jobs = [dict(size=(160, 224), dtype="bf16", B=3), dict(size=(96, 128), dtype="fp32", B=2)]
    for j in jobs:
        j["x"] = rng.integers(0, 256, (j["B"],) + j["size"] + (3,), dtype=np.uint8)

    def work(j, out, iters, phase=0):
        eng = cfa.Engine(j["size"][0], j["size"][1], max_batch=j["B"], dtype=j["dtype"])
        res = []
        for it in range(iters):
            eng.forward_enqueue(j["x"])
            d, l, i = eng.decode_topk(30)
            t = eng.decode_threshold(0.3, 0.3, 256)
            if (it + phase) % 3 == 0:
                # same weights again: graphs dropped (re-captured two forwards later), results unchanged.  The uploads of one
                # thread run while the other thread captures a forward graph of its context: neither may touch the legacy
                # stream (a legacy-stream hipMemcpy failed here with "would make the legacy stream depend on a capturing
                # blocking stream" when the two happened to meet)
                eng.load_state_dict(cfa.weights.synthetic_state_dict(0))
            res.append((d, l, i, [a for a, _ in t]))
        eng.close()
        out.append(res)

    ref = []
    for j in jobs:
        o = []; work(j, o, 2); ref.append(o[0][0])
    outs = [[], []]
    th = [threading.Thread(target=work, args=(jobs[k], outs[k], 12, k)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for k in range(2):
        assert len(outs[k]) == 1 and len(outs[k][0]) == 12
        for d, l, i, t in outs[k][0]:
            assert np.array_equal(d, ref[k][0]) and np.array_equal(l, ref[k][1]) and np.array_equal(i, ref[k][2])
            assert all(np.array_equal(a, b) for a, b in zip(t, ref[k][3]))

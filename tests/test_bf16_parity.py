"""Parity of the BENCHMARKED mode (bf16 storage, fused kernels) against a bf16-emulating oracle.

``oracle/bf16_emulation.py`` is the fp32 oracle with every value rounded at exactly the engine's storage
points (bf16 for HBM tensors and the project operand, fp16 round-toward-zero for the expanded tile, fp16 taps,
pre-scaled Swish weights, collapsed heads); it is pinned to the reference by a hooked run of the reference's own
module graph (tests/golden/net_bf16emu.npz, tools/gen_goldens_bf16emu.py).  What is left between the GPU and
that emulation is fp32 summation order and the 1-ulp exp/rcp approximations, i.e. ~1e-7 relative -- visible only
as occasional 1-ulp rounding flips at the next storage point.  Bound used per kernel:

    tol = 2^-7 |emu| + 2^-8 rms(emu)                   (one bf16 ulp at the output + flip noise from inside)
    accept: no element beyond 2 tol, <= 1e-5 of them beyond tol, >= 99 % of a bf16 output BIT-IDENTICAL

Measured on MI355X (profiles/r02_bf16_parity_stats.md): 99.75-99.99 % of every kernel's outputs are bit-identical
to the emulation, worst element 1.25 tol.  A wrong depthwise tap on one pixel moves its outputs by ~2^-3 rms
(32 tol); on one edge column of a tile it also breaks the bit-identity fraction.

End to end the two drift apart (a quantised 16-block network is chaotic at the 1-ulp level: each flip
perturbs the next block's sums by far more than 1e-7, so after a few blocks every element is "re-rolled");
that is a property of bf16 storage, not of the kernels -- the hooked reference and the emulation show the
same drift between each other on CPU (tools/gen_goldens_bf16emu.py prints it).  So the engine is checked
LAYER BY LAYER at production shape with teacher forcing: every kernel's output on the GPU is compared with
the emulation of that kernel applied to the GPU's own input(s) for it (``cf_forward_trace``).
"""
import numpy as np
import pytest
import torch

import centerface_amd as cfa
from centerface_amd import ops
from oracle import bf16_emulation as E
from oracle import centerface_oracle as O

pytestmark = pytest.mark.gpu

SD = cfa.weights.synthetic_state_dict(0)


def _bf16_normal(rng, shape, scale=1.0):
    return E.q_bf16(torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))).numpy()


def _ratio(got, ref):
    ref = np.asarray(ref, np.float64)
    return np.abs(np.asarray(got, np.float64) - ref) / E.tolerance(ref)


def _assert_close(got, ref, what, bf16_output=True):
    ref = ref.numpy() if hasattr(ref, "numpy") else ref
    r = _ratio(got, ref)
    stat = (float(r.max()), float((r > 1).mean()), float((r > 0.5).mean()), float((r > 0).mean()))
    assert E.accept(stat, bf16_output, r.size), "%s: max |d|/tol %.2f at %s, frac > tol %.1e, > tol/2 %.1e, differing %.1e" % (
        (what, stat[0], np.unravel_index(r.argmax(), r.shape)) + stat[1:])
    return stat


def _block_shapes(size=640):
    """(prefix, cin, cout, k, s, H_in) of the 11 MBConv blocks behind the fused stem at a given input size."""
    out, h = [], size // 2                      # the stem halves the map; layer0.0 (inside the fused stem) keeps it
    for prefix, cin, cout, t, k, s in O.blocks_table():
        if prefix == "layer0.0":
            continue
        out.append((prefix, cin, cout, k, s, h))
        h //= s
    return out


BLOCKS = _block_shapes()


def _w(prefix):
    we, wd, wp = (SD["%s.conv.%s.weight" % (prefix, j)] for j in ("0.1", "1.1", "2"))
    return we.reshape(we.shape[0], -1), wd, wp.reshape(wp.shape[0], -1)


# ------------------------------------------------------------------------------- every production instance, production shape
@pytest.mark.parametrize("blk", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_production_mbconv_instances_vs_emulation(blk):
    """Every MBConv template instance the bf16 engine launches at 640x640 (the op-level entry points go through
    the same geometry tables as the engine: ``mbconv_px_kernel<3,2,1,false,4,1,32,8,16>`` for layer1.0, ...,
    ``expdw_px_kernel`` + ``pw_wlds_kernel`` for layer4.0-6.0), at its production map size with the production
    weights, B = 2, against the emulation at one bf16 ulp + flip noise."""
    prefix, cin, cout, k, s, h = blk
    rng = np.random.default_rng(sum(ord(ch) for ch in prefix))
    x = _bf16_normal(rng, (2, cin, h, h), 1.2)
    we, wd, wp = _w(prefix)
    res = cin == cout and s == 1
    if prefix in E.SPLIT_BLOCKS:
        y = ops.expand_dw(x, we, wd, k, s, dtype="bf16")
        _assert_close(y, E.expand_dw(torch.from_numpy(x), we, wd, k, s, out_scaled=False), prefix + " expand+dw")
        # the project GEMM on the GPU's own depthwise output: the only error left is the output rounding
        o = ops.conv_pw(y, wp, residual=x if res else None, dtype="bf16")
        _assert_close(o, E.pw_op(y, wp, residual=x if res else None), prefix + " project")
    else:
        y = ops.mbconv(x, we, wd, wp, k, s, dtype="bf16")
        _assert_close(y, E.mbconv_fused(torch.from_numpy(x), we, wd, wp, k, s, res), prefix)


@pytest.mark.parametrize("prefix", E.SPLIT_BLOCKS)
def test_production_project_gemm_at_batch64(prefix):
    """``pw_wlds_kernel`` picks its LDS ring depth from the grid size, so the B = 64 instance (2 stages) differs
    from the B = 2 one (3-4 stages): the late project GEMMs at the benchmark's batch, 40x40 / 20x20 maps."""
    _, cin, cout, k, s, h = [b for b in BLOCKS if b[0] == prefix][0]
    ho = h // s
    rng = np.random.default_rng(cout)
    we, wd, wp = _w(prefix)
    y = _bf16_normal(rng, (64, we.shape[0], ho, ho), 0.7)
    res = _bf16_normal(rng, (64, cout, ho, ho), 1.0) if (cin == cout and s == 1) else None
    o = ops.conv_pw(y, wp, residual=res, dtype="bf16")
    _assert_close(o, E.pw_op(y, wp, residual=res), prefix + " project B=64")


@pytest.mark.parametrize("prefix", ["layer5.0", "layer5.1", "layer6.0"])
def test_project_gemm_k_split_instances_at_1280_maps(prefix):
    """On the 40x40 late maps of 1280x1280 inputs (BASELINE configs[4], four images per GPU) the project GEMMs run as
    ``pw_ksplit_kernel`` (K split over the four waves of a workgroup, partial sums added in wave order): against the
    emulation, and -- the kernel is chosen by the LAYER's map size, not by the batch -- an image's rows must not change
    with the batch it travels in."""
    _, cin, cout, k, s, h = [b for b in BLOCKS if b[0] == prefix][0]
    ho = 2 * (h // s)                                      # the 1280x1280 instance of the layer
    rng = np.random.default_rng(cout + 1)
    we, wd, wp = _w(prefix)
    y = _bf16_normal(rng, (4, we.shape[0], ho, ho), 0.7)
    res = _bf16_normal(rng, (4, cout, ho, ho), 1.0) if (cin == cout and s == 1) else None
    o = ops.conv_pw(y, wp, residual=res, dtype="bf16")
    _assert_close(o, E.pw_op(y, wp, residual=res), prefix + " project, 40x40 map, B=4")
    o1 = ops.conv_pw(y[2:3], wp, residual=None if res is None else res[2:3], dtype="bf16")
    assert np.array_equal(o1[0], o[2])


# ------------------------------------------------------------------------------- the engine itself, layer by layer
def _trace_all(eng, x):
    plan = eng.plan()
    rec = {}
    for op in plan:
        if not op["fused_away"]:
            rec[op["name"]] = eng.trace(x, op["index"])
    return plan, rec


def _engine_record(eng, x):
    """Layer trace of the engine re-keyed to the emulation's block names (+ the fp32 head maps)."""
    plan, rec = _trace_all(eng, x)
    g = {("img_u8" if x.dtype == np.uint8 else "x"): x}
    for name, v in rec.items():
        if name == "first_conv+layer0.0":
            g["layer0.0"] = v
        elif name.endswith(".mbconv") or name.endswith(".project"):
            g[name.rsplit(".", 1)[0]] = v
        elif name.endswith(".expand+dw"):
            g[name.rsplit(".", 1)[0] + ".dw"] = v
        elif name in ("conv_last", "up1", "up2", "up3"):
            g[name] = v
        elif name == "conv_last+up1+up2":              # the fused neck launch: only its last map reaches HBM
            g["up2"] = v
        elif name in ("up3+heads", "heads"):
            g["hm"], g["wh"], g["lm"], g["reg"] = v[:, 15:16], v[:, 1:3], v[:, 3:13], v[:, 13:15]
    return g, rec


@pytest.mark.parametrize("size,B", [((640, 640), 2), ((480, 640), 2), ((96, 128), 3), ((160, 224), 2), ((64, 96), 2), ((32, 32), 1), ((32, 640), 2)])
def test_bf16_engine_layer_by_layer_teacher_forced(size, B):
    """The bf16 ENGINE (its real launch plan, buffers and kernels) at the production 640x640 shape, a VGA
    bucket and sizes that are not a multiple of any tile (every kernel has edge tiles; maps smaller than a
    tile): each plan entry's output against the emulation of that entry applied to the engine's own inputs
    for it.  19 launches, each within one bf16 ulp + flip noise -- end to end there is no tighter statement
    to make about a bf16 network than this (see the module docstring)."""
    H, W = size
    rng = np.random.default_rng(H + 3 * W)
    x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    eng = cfa.Engine(H, W, max_batch=B, dtype="bf16")
    g, rec = _engine_record(eng, x)
    if "conv_last" not in g:
        # conv_last and up1 live only in LDS inside the fused neck launch: take them from the three-kernel plan and hold the
        # fused launch to bit-equality with that plan's up2 (tests/test_gpu_parity.py does the same on more shapes)
        e3 = cfa.Engine(H, W, max_batch=B, dtype="bf16", neck=False)
        g3, _ = _engine_record(e3, x)
        assert np.array_equal(g3["up2"], g["up2"]) and np.array_equal(g3["layer6.0"], g["layer6.0"])
        g["conv_last"], g["up1"] = g3["conv_last"], g3["up1"]
        e3.close()
    stats = E.check_blockwise(SD, g, detail=True)
    assert len(stats) == 11 + 1 + 3 + 4 + (1 if "up3" in g else 0) + sum(k.endswith(".dw") for k in g), sorted(stats)
    bad = {k: v for k, v in stats.items() if not E.accept(v, bf16_output=not k.startswith("head."))}
    assert not bad, bad
    # the split blocks' expand+depthwise launches (their output is an engine buffer too)
    for prefix in E.SPLIT_BLOCKS:
        _, cin, cout, k, s, _ = [b for b in BLOCKS if b[0] == prefix][0]
        we, wd, wp = _w(prefix)
        names = [b[0] for b in BLOCKS]
        prev = names[names.index(prefix) - 1]
        emu = E.expand_dw(torch.from_numpy(g[prev]), we, wd, k, s, out_scaled=False)
        _assert_close(rec[prefix + ".expand+dw"], emu, prefix + ".expand+dw")
    # and the full forward (graph replay path) reproduces the traced heads bit for bit
    eng.forward_enqueue(x); eng.forward_enqueue(x)
    hd = eng.heads()
    for k in ("hm", "wh", "lm", "reg"):
        assert np.array_equal(hd[k], g[k]), k
    eng.close()


def test_float_input_path_layer0_vs_emulation(golden):
    """CF_IN_F32_NCHW staging (already-normalised tensor, as the reference hands it to net()) through the fused stem."""
    g = golden("net_bf16emu")
    x = g["x_c"]
    eng = cfa.Engine(x.shape[2], x.shape[3], max_batch=1, dtype="bf16")
    got = eng.trace(x, 0)
    _assert_close(got, E.from_bf16_bits(g["layer0.0_c"]), "stem0 (f32 input) vs hooked reference")
    eng.close()


# ------------------------------------------------------------------------------- end to end
def _certified_ranks(hm_sig, K, e):
    """Ranks of the emulated top-K whose cell index is determined whatever a perturbation of every heat-map cell
    by at most ``e`` does: the cell is a 3x3 peak by a margin > 2e, no other potential peak (cell within 2e of
    its 3x3 maximum) has a score within 2e of it, and every potential peak above it is a definite one."""
    h, w = hm_sig.shape
    pad = np.full((h + 2, w + 2), -np.inf, np.float32)
    pad[1:-1, 1:-1] = hm_sig
    nb = np.full((h, w), -np.inf, np.float32)
    for dy in range(3):
        for dx in range(3):
            if dy == 1 and dx == 1:
                continue
            nb = np.maximum(nb, pad[dy:dy + h, dx:dx + w])
    margin = (hm_sig - nb).ravel()
    v = hm_sig.ravel()
    potential = margin >= -2 * e
    definite = margin > 2 * e
    order = np.argsort(-np.where(margin >= 0, v, 0), kind="stable")[:K]      # the emulated top-K (peaks only)
    pot_idx = np.nonzero(potential)[0]
    pot_v = v[pot_idx]
    cert = np.zeros(K, bool)
    for r, c in enumerate(order):
        if not definite[c]:
            continue
        near = np.abs(pot_v - v[c]) <= 2 * e
        near[pot_idx == c] = False
        above_ok = definite[pot_idx[pot_v > v[c] + 2 * e]].all()
        cert[r] = (not near.any()) and above_ok
    return order, cert


def test_batch64_end_to_end_vs_emulation():
    """BASELINE configs[1] (B = 64, 640x640, bf16, top-100): head maps, scores, boxes and indices of the engine
    against the emulating oracle on four images spread over the batch (every image runs the same kernels, the
    batch index is only blockIdx.z; the CPU emulation of all 64 would take minutes).  Stated bounds, with the
    measured values in the assertion messages:
      head maps            mean |d| <= 0.006 rms(map), max |d| <= 0.12 rms(map)   (flip drift, see module docstring)
      sigmoid(hm) scores   max |d| <= 0.02
      top-100 indices      at EVERY rank: identical to the emulation's index, or a near-tie -- the engine's cell at
                           that rank is a 3x3 peak of the emulated map up to 2e and its emulated score is within 2e of
                           the emulated score at that rank (e = the measured max score error of that image);
                           identical at every rank the emulation CERTIFIES against any perturbation <= e of every
                           cell (few ranks: top-100 score gaps are ~0.005, the same size as e); set overlap >= 97 / 100
      boxes                for identical indices: |d| <= 0.05 map pixels."""
    B, S, K = 64, 640, 100
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    eng = cfa.Engine(S, S, max_batch=B, dtype="bf16")
    eng.forward_enqueue(imgs)
    hd = eng.heads(sigmoid_hm=True)
    dets, lms, inds = eng.decode_topk(K=K)
    # the decode kernel is exact on the engine's own maps (bit-exact vs the oracle's ctdet_decode)
    pick = [0, 21, 42, 63]
    rd, rl, ri = O.ctdet_decode(hd["hm_sigmoid"][pick], hd["wh"][pick], hd["reg"][pick], K, hd["lm"][pick])
    assert np.array_equal(inds[pick], ri) and np.array_equal(dets[pick], rd) and np.array_equal(lms[pick], rl)
    n_cert = n_same = 0
    for j, b in enumerate(pick):
        emu = E.forward(SD, img_u8=imgs[b:b + 1])
        for k in ("hm", "wh", "lm", "reg"):
            ref = emu[k].numpy()
            d = np.abs(hd[k][b:b + 1] - ref)
            rms = float(np.sqrt((ref ** 2).mean()))
            assert d.mean() <= 0.006 * rms and d.max() <= 0.12 * rms, (b, k, float(d.mean()) / rms, float(d.max()) / rms)
        sg = O.sigmoid_clamp(emu["hm"]).numpy()[0, 0]
        e = float(np.abs(hd["hm_sigmoid"][b, 0] - sg).max())
        assert e <= 0.02, (b, e)
        order, cert = _certified_ranks(sg, K, e)
        n_cert += int(cert.sum())
        assert np.array_equal(inds[b][cert], order[cert]), (b, int(cert.sum()))
        same = inds[b] == order
        n_same += int(same.sum())
        flat = sg.ravel()
        pad = np.pad(sg, 1, constant_values=-np.inf)
        nbmax = np.max([pad[dy:dy + sg.shape[0], dx:dx + sg.shape[1]] for dy in range(3) for dx in range(3)], axis=0).ravel()
        for r in np.nonzero(~same)[0]:
            c = inds[b][r]
            assert abs(flat[c] - flat[order[r]]) <= 2 * e and flat[c] >= nbmax[c] - 2 * e, (b, int(r), float(flat[c]), float(flat[order[r]]), e)
        assert len(set(inds[b].tolist()) & set(order.tolist())) >= 97, b
        ed, _, _ = O.ctdet_decode(sg[None, None], emu["wh"].numpy(), emu["reg"].numpy(), K)
        assert np.abs(dets[b][same][:, :4] - ed[0][same][:, :4]).max() <= 0.05
        assert np.abs(dets[b][same][:, 4] - ed[0][same][:, 4]).max() <= e + 1e-7
    # (measured: ~45 % of the ranks carry the identical index -- the synthetic weights give ~250 cells above 0.3 whose
    #  top-100 scores are ~0.005 apart, the size of e, so neighbouring ranks swap; every swap was checked above)
    print("certified ranks %d / 400, identical ranks %d / 400" % (n_cert, n_same))
    eng.close()


# ------------------------------------------------------------------------------- reload
def test_weight_reload_after_graph_capture():
    """cf_load_weights on a live context drops the captured hipGraphs (they hold the old weight pointers) and
    frees the old weight set: outputs after a reload equal those of a fresh no-graph context."""
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, (2, 64, 96, 3), dtype=np.uint8)
    sd2 = cfa.weights.synthetic_state_dict(7)
    for dtype in ("bf16", "fp32"):
        eng = cfa.Engine(64, 96, max_batch=2, dtype=dtype, weights=SD)
        for _ in range(3):
            eng.forward_enqueue(x)                       # first: eager, second: capture, third: replay
        a = eng.heads()
        assert eng.graph_stats()[0] >= 1
        eng.load_state_dict(sd2)
        assert eng.graph_stats()[0] == 0
        for _ in range(3):
            eng.forward_enqueue(x)
        b = eng.heads()
        fresh = cfa.Engine(64, 96, max_batch=2, dtype=dtype, weights=sd2, graph=False)
        fresh.forward_enqueue(x)
        c = fresh.heads()
        for k in ("hm", "wh", "lm", "reg"):
            assert np.array_equal(b[k], c[k]), (dtype, k)
            assert not np.array_equal(a[k], b[k]), (dtype, k)
        eng.close(); fresh.close()

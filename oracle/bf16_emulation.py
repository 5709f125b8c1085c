"""ORACLE -- test infrastructure only.  CPU emulation of the product's bf16 THROUGHPUT mode.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s parity/cpu_baseline legs may import this
file; the product (``lightweight-face-detection-centernet_amd/``) never does.

What it is: the fp32 oracle of ``centerface_oracle.py`` (the reference's network, model/centernet.py:
58-70,89-140,179-204,240-280) with every value rounded at exactly the points where the bf16 engine
(``CF_BF16``, default flags: fused blocks, collapsed heads, fused up3+heads) stores or re-quantises it:

  =====================================  =====================================================================
  storage point (product file)           rounding emulated here
  =====================================  =====================================================================
  every HBM activation tensor            bf16, round-to-nearest-even (``v_cvt_pk_bf16_f32``, cf_common.h)
  network input (cf_stem0.hip)           u8: fma(u, 1/(255 std), -mean/std) in fp32 -> bf16; f32 NCHW: -> bf16
  expand / stem weights                  bf16(-log2(e) * w)   (Swish pre-scale folded in, cf_mbconv2.hip:775)
  expanded tile E in LDS                 fp16, round-TOWARD-ZERO, saturating (``v_cvt_pkrtz_f16_f32``)
                                         of  u / (1 + 2^u),  u = -log2(e) * expand output
  depthwise taps                         fp16, round-to-nearest-even, saturating (host packer)
  depthwise accumulate                   fp32 (``v_dot2c_f32_f16``)
  project operand (fused blocks)         bf16 of  d / (1 + 2^d)  (d = pre-scaled depthwise sum)
  project weights (fused blocks)         bf16(-ln(2) * w)
  layer4.0-6.0 depthwise output (HBM)    bf16 of  (d / (1 + 2^d)) * -ln(2); project weights plain bf16(w)
  conv_last / IDAUp 1x1 weights          bf16(float(w * bn_scale)), bn_scale and shift folded in float64
  IDAUp deconv taps, all biases          fp32
  head weights                           collapsed (W = w1 . w0 in float64) -> float -> bf16; bias fp32
  heads output                           fp32 (not re-quantised)
  =====================================  =====================================================================

Accumulation is fp32 everywhere (MFMA bf16 -> fp32), as here (torch CPU conv on fp32 tensors whose values
are exactly representable in the emulated formats, so products are exact and only the fp32 summation
order differs).  What this emulation does NOT reproduce bit for bit: fp32 summation order, the 1-ulp
``v_exp_f32`` / ``v_rcp_f32`` approximations, fma contraction.  Those are ~1e-7 relative; they matter
only when they push a value across a rounding boundary of the next storage point (a 1-ulp "flip":
2^-8 relative for bf16, 2^-11 for fp16), which happens for ~1e-3 of the elements.  The GPU tests
therefore compare at  |got - emu| <= 2^-7 |emu| + 2^-8 rms(emu)  per op (one bf16 ulp at the output plus
flip noise from the intermediates) -- against 6e-2 / "mean < 0.05" when comparing with the fp32 oracle.

Parity status: PINNED to the reference.  ``tools/gen_goldens_bf16emu.py`` runs the REFERENCE's own module
graph (``model.centernet.efficientnet_b0``) with the same quantised weights and with forward hooks that
insert the same activation roundings, and stores its outputs in ``tests/golden/net_bf16emu.npz``;
``tests/test_oracle_vs_golden.py::test_bf16_emulation_matches_hooked_reference`` checks this file against
them.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import centerface_oracle as O

NEG_LOG2E = np.float32(-1.44269504088896341)
NEG_LN2 = np.float32(-0.69314718055994531)


# ----------------------------------------------------------------------------- rounding primitives
def q_bf16(t):
    """fp32 -> bf16 (RNE) -> fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


def q_f16_rne_sat(t):
    """fp32 -> fp16 round-to-nearest-even, saturating at +-65504 (host packer host_f32_to_f16) -> fp32."""
    return torch.clamp(t, -65504.0, 65504.0).to(torch.float16).to(torch.float32)


def q_f16_rtz_sat(t):
    """fp32 -> fp16 round-toward-zero, saturating (v_cvt_pkrtz_f16_f32) -> fp32."""
    t = torch.clamp(t, -65504.0, 65504.0)
    a = t.abs()
    bits = a.contiguous().view(torch.int32)
    normal = (bits & ~0x1FFF).view(torch.float32)                       # drop 13 mantissa bits
    sub = torch.floor(a * (2.0 ** 24)) * (2.0 ** -24)                     # fp16 subnormal grid
    r = torch.where(a < 2.0 ** -14, sub, normal)
    return torch.copysign(r, t)


def swish_prescaled(u):
    """u / (1 + 2^u): with u = -log2(e) x this is -log2(e) * swish(x) (cf_mbconv2.hip swish2_prescaled)."""
    return u / (1.0 + torch.exp2(u))


def swish_exp2(x):
    """cf_common.h swish2: x / (1 + 2^(-log2(e) x))."""
    return x / (1.0 + torch.exp2(x * float(NEG_LOG2E)))


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))


def _f32mul(c, w):
    """float32 product of a float32 constant and a float32 tensor (what the host packers compute)."""
    return (_t(w).to(torch.float32) * torch.tensor(float(c), dtype=torch.float32))


# ----------------------------------------------------------------------------- blocks
def normalise_u8(img_u8):
    """cf_stem0.hip:379-411: one fp32 fma per byte, then bf16.  img uint8 [B,H,W,3] BGR -> [B,3,H,W]."""
    f = np.float32
    std = [f(0.289), f(0.274), f(0.278)]
    mean = [f(0.408), f(0.447), f(0.470)]
    sc = np.array([f(1.0) / (f(255.0) * s) for s in std], dtype=np.float32)
    sh = np.array([-m / s for m, s in zip(mean, std)], dtype=np.float32)
    x = (img_u8.astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)).astype(np.float32)   # fma: one rounding
    return q_bf16(torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2))))


def expand_dw(x, we, wd, k, s, out_scaled):
    """expand 1x1 + Swish -> fp16 tile -> depthwise k x k + Swish (model/centernet.py:109-114) as
    cf_mbconv2.hip computes it.  Returns the pre-scaled depthwise activation  -log2(e) * swish(dw)  when
    ``out_scaled`` (the fused blocks' project operand) else the true-scale one (layer5.0-6.0), both bf16."""
    hid = we.shape[0]
    weq = q_bf16(_f32mul(NEG_LOG2E, we)).reshape(hid, -1, 1, 1)
    u = F.conv2d(x, weq)
    e = q_f16_rtz_sat(swish_prescaled(u))
    wdq = q_f16_rne_sat(_t(wd).to(torch.float32)).reshape(hid, 1, k, k)
    d = F.conv2d(O.same_pad(e, k, s), wdq, None, s, 0, 1, hid)
    y = swish_prescaled(d)
    if not out_scaled:
        y = y * float(NEG_LN2)
    return q_bf16(y)


def dw_only(e_scaled, wd, k, s):
    """layer0.0 inside the fused stem kernel: the tile is the stem output (already fp16, pre-scaled)."""
    c = e_scaled.shape[1]
    wdq = q_f16_rne_sat(_t(wd).to(torch.float32)).reshape(c, 1, k, k)
    d = F.conv2d(O.same_pad(e_scaled, k, s), wdq, None, s, 0, 1, c)
    return q_bf16(swish_prescaled(d))


def mbconv_fused(x, we, wd, wp, k, s, residual):
    """MBConvBlock.forward (model/centernet.py:89-140), fused kernel mbconv_px_kernel (layer1.0-3.1)."""
    y = expand_dw(x, we, wd, k, s, out_scaled=True)
    wpq = q_bf16(_f32mul(NEG_LN2, wp)).reshape(wp.shape[0], -1, 1, 1)
    o = F.conv2d(y, wpq)
    if residual:
        o = x + o
    return q_bf16(o)


def mbconv_split(x, we, wd, wp, k, s, residual):
    """Same block as expdw_px_kernel + pw_wlds_kernel (layer4.0 - 6.0): depthwise output in HBM."""
    y = expand_dw(x, we, wd, k, s, out_scaled=False)
    wpq = q_bf16(_t(wp).to(torch.float32)).reshape(wp.shape[0], -1, 1, 1)
    o = F.conv2d(y, wpq)
    if residual:
        o = o + x
    return q_bf16(o)


def stem0(x, ws, wd, wp):
    """first_conv + layer0.0 (model/centernet.py:224,213) as stem0_px_kernel: x bf16-valued [B,3,H,W]."""
    wsq = q_bf16(_f32mul(NEG_LOG2E, ws))
    u = F.conv2d(O.same_pad(x, 3, 2), wsq, None, 2)
    e = q_f16_rtz_sat(swish_prescaled(u))
    y = dw_only(e, wd, 3, 1)
    wpq = q_bf16(_f32mul(NEG_LN2, wp)).reshape(wp.shape[0], -1, 1, 1)
    return q_bf16(F.conv2d(y, wpq))


def bn_fold(sd, prefix, eps):
    """float64 fold of eval-mode BatchNorm into (scale, shift) -- cf_runtime.hip bn_fold."""
    g, b = (_t(sd[prefix + k]).double() for k in (".weight", ".bias"))
    mu, var = (_t(sd[prefix + k]).double() for k in (".running_mean", ".running_var"))
    scale = g / torch.sqrt(var + float(np.float32(eps)))
    return scale, b - mu * scale


def folded_pw(sd, wkey, bnkey, eps):
    w = _t(sd[wkey]).double()
    scale, shift = bn_fold(sd, bnkey, eps)
    wq = q_bf16((w * scale.reshape(-1, 1, 1, 1)).float())
    return wq, shift.float()


def conv_last(x, sd):
    """conv_1x1_bn (model/centernet.py:179-184) as pw_kernel<bf16, ACT=1, BIAS>."""
    wq, b = folded_pw(sd, "conv_last.0.weight", "conv_last.1", 1e-5)
    return q_bf16(swish_exp2(F.conv2d(x, wq, b)))


def idaup(low, skip, sd, prefix):
    """IDAUp.forward (model/centernet.py:200-204) as the pw_kernel IDAUp epilogue / uphead phase A."""
    wq, b = folded_pw(sd, prefix + ".conv.0.weight", prefix + ".conv.1", 1e-3)
    scale, shift = bn_fold(sd, prefix + ".bn_up", 1e-3)
    c = low.shape[1]
    upw = (_t(sd[prefix + ".up.weight"]).double() * scale.reshape(-1, 1, 1, 1)).float()    # [C,1,2,2]
    up = F.conv_transpose2d(low, upw, None, 2, 0, 0, c) + shift.float().reshape(1, -1, 1, 1)
    return q_bf16(F.relu(F.conv2d(skip, wq, b)) + F.relu(up))


_HEAD_OF_OUT = (0, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3)


def collapsed_head_weights(sd):
    """head_pack_weights(collapsed=1): W[o] = sum_c w1[o,c] w0[head(o),c] in float64 -> float -> bf16."""
    names = O._HEADS
    w1 = torch.cat([_t(sd[n + ".1.weight"]).double().reshape(-1, 24) for n in names])      # [15,24]
    b1 = torch.cat([_t(sd[n + ".1.bias"]).double() for n in names])
    W = torch.zeros(15, 24, 3, 3, dtype=torch.float64)
    bq = torch.zeros(15, dtype=torch.float64)
    for o, hd in enumerate(_HEAD_OF_OUT):
        w0 = _t(sd[names[hd] + ".0.weight"]).double()                                      # [24,24,3,3]
        b0 = _t(sd[names[hd] + ".0.bias"]).double()
        W[o] = torch.einsum("c,cikl->ikl", w1[o], w0)
        bq[o] = b1[o] + (w1[o] * b0).sum()
    return q_bf16(W.float()), bq.float()


def heads(x, sd):
    """The four heads (model/centernet.py:247-261,277-279), collapsed, fp32 out: dict hm/wh/lm/reg."""
    W, b = collapsed_head_weights(sd)
    out = F.conv2d(x, W, b, 1, 1)
    return OrderedDict((("hm", out[:, 0:1]), ("wh", out[:, 1:3]), ("lm", out[:, 3:13]), ("reg", out[:, 13:15])))


SPLIT_BLOCKS = ("layer4.0", "layer4.1", "layer5.0", "layer5.1", "layer6.0")   # expand+dw kernel + project GEMM in the engine


@torch.no_grad()
def forward(sd, x=None, img_u8=None, return_features=False):
    """EfficientNet.forward (model/centernet.py:263-280) as the bf16 engine computes it.  Give either ``x``
    (float32 [B,3,H,W], already normalised: the CF_IN_F32_NCHW input) or ``img_u8`` ([B,H,W,3] BGR)."""
    sd = O.to_torch_sd(sd)
    x = normalise_u8(img_u8) if img_u8 is not None else q_bf16(_t(x).float())
    feats = OrderedDict()
    x = stem0(x, sd["first_conv.0.1.weight"], sd["layer0.0.conv.0.1.weight"], sd["layer0.0.conv.1.weight"])
    feats["layer0.0"] = x
    skips = {}
    for prefix, cin, cout, t, k, s in O.blocks_table():
        if prefix == "layer0.0":
            continue
        we, wd, wp = (sd["%s.conv.%s.weight" % (prefix, j)] for j in ("0.1", "1.1", "2"))
        res = cin == cout and s == 1
        fn = mbconv_split if prefix in SPLIT_BLOCKS else mbconv_fused
        x = fn(x, we.reshape(we.shape[0], -1), wd, wp.reshape(wp.shape[0], -1), k, s, res)
        feats[prefix] = x
        if prefix in ("layer1.1", "layer2.1", "layer4.1"):
            skips[prefix] = x
    x = conv_last(x, sd)
    feats["conv_last"] = x
    x = idaup(x, skips["layer4.1"], sd, "up1"); feats["up1"] = x
    x = idaup(x, skips["layer2.1"], sd, "up2"); feats["up2"] = x
    x = idaup(x, skips["layer1.1"], sd, "up3"); feats["up3"] = x
    out = heads(x, sd)
    return (out, feats) if return_features else out


def tolerance(ref, rel=2.0 ** -7, flip=2.0 ** -8):
    """Per-element bound used by the GPU parity tests: one bf16 ulp of the output + flip noise."""
    ref = np.asarray(ref, np.float64)
    return rel * np.abs(ref) + flip * math.sqrt(float(np.mean(ref * ref)) + 1e-30)


def from_bf16_bits(a):
    """uint16 raw bf16 bits (as stored in tests/golden/net_bf16emu.npz) -> float32 tensor."""
    return torch.from_numpy((np.asarray(a).astype(np.uint32) << 16).view(np.float32))


@torch.no_grad()
def check_blockwise(sd, g, rel=2.0 ** -7, flip=2.0 ** -8, detail=False):
    """Teacher-forced check of every block of this emulation against a record ``g`` of block outputs produced by
    someone else (the hooked reference: tests/golden/net_bf16emu.npz; or the GPU engine's layer trace): each block
    is evaluated on the RECORD's input(s) for it and compared with the record's output.  ``g``: dict with 'x'
    (float32 network input) or 'img_u8', 'layerL.i', 'conv_last', 'up1'..'up3' (float32 or raw bf16 bits) and
    optionally 'hm','wh','lm','reg'.  Returns {block: max over elements of |d| / tolerance} (all <= 1 = pass)."""
    sd = O.to_torch_sd(sd)

    def rec(k):
        v = g[k]
        return from_bf16_bits(v) if np.asarray(v).dtype == np.uint16 else _t(np.asarray(v, np.float32))

    def ratio(got, ref):
        ref = ref.numpy().astype(np.float64)
        r = np.abs(got.numpy().astype(np.float64) - ref) / tolerance(ref, rel, flip)
        if detail:                          # (max, fraction beyond the bound, beyond half of it, differing at all, n)
            return (float(r.max()), float((r > 1).mean()), float((r > 0.5).mean()), float((r > 0).mean()), int(r.size))
        return float(r.max())

    worst = OrderedDict()
    x0 = normalise_u8(np.asarray(g["img_u8"])) if "img_u8" in g else q_bf16(_t(np.asarray(g["x"], np.float32)))
    worst["layer0.0"] = ratio(stem0(x0, sd["first_conv.0.1.weight"], sd["layer0.0.conv.0.1.weight"], sd["layer0.0.conv.1.weight"]),
                              rec("layer0.0"))
    prev = "layer0.0"
    for prefix, cin, cout, t, k, s in O.blocks_table():
        if prefix == "layer0.0":
            continue
        we, wd, wp = (sd["%s.conv.%s.weight" % (prefix, j)] for j in ("0.1", "1.1", "2"))
        we2, wp2, res = we.reshape(we.shape[0], -1), wp.reshape(wp.shape[0], -1), cin == cout and s == 1
        if prefix in SPLIT_BLOCKS and (prefix + ".dw") in g:
            # the record holds the depthwise tensor between the two launches of a split block: teacher-force each launch
            # on its own (expand+depthwise on the block input, project GEMM on the RECORD's depthwise tensor)
            worst[prefix + ".dw"] = ratio(expand_dw(rec(prev), we2, wd, k, s, out_scaled=False), rec(prefix + ".dw"))
            y = pw_op(rec(prefix + ".dw"), wp2, residual=rec(prev) if res else None)
        else:
            fn = mbconv_split if prefix in SPLIT_BLOCKS else mbconv_fused
            y = fn(rec(prev), we2, wd, wp2, k, s, res)
        worst[prefix] = ratio(y, rec(prefix))
        prev = prefix
    worst["conv_last"] = ratio(conv_last(rec("layer6.0"), sd), rec("conv_last"))
    worst["up1"] = ratio(idaup(rec("conv_last"), rec("layer4.1"), sd, "up1"), rec("up1"))
    worst["up2"] = ratio(idaup(rec("up1"), rec("layer2.1"), sd, "up2"), rec("up2"))
    if "up3" in g:
        worst["up3"] = ratio(idaup(rec("up2"), rec("layer1.1"), sd, "up3"), rec("up3"))
        up3 = rec("up3")
    else:                                   # the engine fuses up3 into the head kernel: its tile never reaches HBM
        up3 = idaup(rec("up2"), rec("layer1.1"), sd, "up3")
    if "hm" in g:
        out = heads(up3, sd)
        for h in O._HEADS:
            worst["head." + h] = ratio(out[h], _t(np.asarray(g[h], np.float32)))
    return worst


# ----------------------------------------------------------------------------- single ops (unfused bf16 kernels)
def pw_op(x, w, bias=None, act="none", residual=None):
    """cf_pw.hip on bf16 storage: bf16 x, bf16(w), fp32 accumulate (+bias) -> act -> (+residual) -> bf16."""
    x = q_bf16(_t(x).float())
    w = _t(w).float()
    y = F.conv2d(x, q_bf16(w.reshape(w.shape[0], -1, 1, 1)), None if bias is None else _t(bias).float())
    y = swish_exp2(y) if act == "swish" else (F.relu(y) if act == "relu" else y)
    if residual is not None:
        y = y + q_bf16(_t(residual).float())
    return q_bf16(y)


def dw_op(x, w, k, s, pad=None, act="swish", bias=None):
    """cf_dw.hip on bf16 storage: bf16 x, fp32 taps, fp32 accumulate (+bias) -> act -> bf16."""
    x = q_bf16(_t(x).float())
    c = x.shape[1]
    if pad is None:
        p = max(k - s, 0)
        pad = (p // 2, p - p // 2)
    y = F.conv2d(F.pad(x, [pad[0], pad[1], pad[0], pad[1]]), _t(w).float().reshape(c, 1, k, k),
                 None if bias is None else _t(bias).float(), s, 0, 1, c)
    return q_bf16(O.swish(y) if act == "swish" else y)


def mbconv_unfused(x, we, wd, wp, k, s, residual):
    """MBConvBlock as three bf16 kernels (CF_FLAG_NO_FUSE / the per-op composition of the golden tests)."""
    y = x
    if we is not None:
        y = pw_op(y, we, act="swish")
    y = dw_op(y, wd, k, s)
    return pw_op(y, wp, residual=x if residual else None)


def accept(stat, bf16_output=True, n=None):
    """The acceptance rule of the parity tests on a (max |d|/tol, frac > tol, frac > tol/2, frac differing) tuple:
    no element beyond 2x the bound, at most 1e-5 of them beyond the bound (rounding flips have a tail: a flipped
    large operand times a large weight), and -- for bf16 outputs -- at least 99 % of the elements BIT-IDENTICAL to
    the emulation (measured: 99.75-99.99 %, profiles/r02_bf16_parity_stats.md).  ``n`` (element count) relaxes the
    fractions to "at most 2 / at most 8 elements" on tensors too small for them to mean anything."""
    mx, f1, _, fd = stat[:4]
    n = stat[4] if len(stat) > 4 else n
    lim1, limd = 1e-5, 1e-2
    if n:
        lim1, limd = max(lim1, 2.0 / n), max(limd, 8.0 / n)
    return mx <= 2.0 and f1 <= lim1 and (fd <= limd or not bf16_output)


def shuffle_v2_block(x, sd, inp, oup, mid, ksize, stride, prefix=""):
    """ShuffleV2Block.forward (model/blocks.py:47-62) as cf_op_shufflev2 computes it on bf16 storage: BN folded in
    float64 into bf16 1x1 weights / fp32 depthwise taps + fp32 bias, every intermediate a bf16 HBM tensor."""
    sd = O.to_torch_sd(sd)
    x = q_bf16(_t(x).float())
    pad = (ksize // 2, ksize // 2)

    def fold(conv, bn):
        scale, shift = bn_fold(sd, prefix + bn, 1e-5)
        w = sd[prefix + conv + ".weight"].double()
        return (w * scale.reshape(-1, 1, 1, 1)).float(), shift.float()

    def main(v):
        w, b = fold("branch_main.0", "branch_main.1")
        v = pw_op(v, w, b, "relu")
        w, b = fold("branch_main.3", "branch_main.4")
        v = dw_op(v, w, ksize, stride, pad, "none", b)
        w, b = fold("branch_main.5", "branch_main.6")
        return pw_op(v, w, b, "relu")
    if stride == 1:
        return torch.cat((x[:, 0::2], main(x[:, 1::2])), 1)
    w, b = fold("branch_proj.0", "branch_proj.1")
    p = dw_op(x, w, ksize, stride, pad, "none", b)
    w, b = fold("branch_proj.2", "branch_proj.3")
    return torch.cat((pw_op(p, w, b, "relu"), main(x)), 1)

"""numpy-in / numpy-out wrappers of the per-op C-ABI entry points (cf_op_*).  Signatures mirror the
torch ops of the reference they replace (NCHW float32 arrays); each call runs the production HIP
kernel on the GPU.  Used by the parity tests; no CPU path."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import f32, ptr

_DT = {"fp32": 0, "bf16": 1, "fp32_split": 2, "bf16x3": 2}


def conv_dw(x, w, k, stride, pad=None, act="swish", bias=None, dtype="fp32", device=0):
    """ZeroPad2d -> depthwise Conv2d -> act (model/centernet.py:58-70; blocks.py:26-29).
    ``pad`` = (lo, hi) applied to both axes; default is the reference's TF-SAME rule."""
    x, w = f32(x), f32(w)
    B, Cc, H, W = x.shape
    if pad is None:
        p = max(k - stride, 0)
        pad = (p // 2, p - p // 2)
    Ho = (H + pad[0] + pad[1] - k) // stride + 1
    Wo = (W + pad[0] + pad[1] - k) // stride + 1
    y = np.empty((B, Cc, Ho, Wo), np.float32)
    _lib.check(_lib.lib().cf_op_dwconv(device, _DT[dtype], ptr(x), ptr(w), ptr(f32(bias)), ptr(y), B, Cc, H, W,
                                       k, stride, pad[0], pad[1], {"none": 0, "swish": 1}[act]), op=True)
    return y


def conv_pw(x, w, act="none", bias=None, residual=None, dtype="fp32", device=0):
    """1x1 Conv2d [+bias] [+act] [+residual] (model/centernet.py:109-110,117-118,134-137)."""
    x = f32(x)
    w = f32(np.asarray(w).reshape(w.shape[0], -1))
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = np.empty((B, Cout, H, W), np.float32)
    _lib.check(_lib.lib().cf_op_pwconv(device, _DT[dtype], ptr(x), ptr(w), ptr(f32(bias)), ptr(f32(residual)),
                                       ptr(y), B, Cin, Cout, H, W, {"none": 0, "swish": 1, "relu": 2}[act]), op=True)
    return y


def mbconv(x, w_exp, w_dw, w_proj, k, stride, dtype="fp32", device=0):
    """MBConvBlock.forward (model/centernet.py:89-140, se=False) as the single fused kernel."""
    x = f32(x)
    B, Cin, H, W = x.shape
    w_exp = f32(np.asarray(w_exp).reshape(w_exp.shape[0], -1))
    hid = w_exp.shape[0]
    w_dw = f32(np.asarray(w_dw).reshape(hid, k * k))
    w_proj = f32(np.asarray(w_proj).reshape(w_proj.shape[0], -1))
    Cout = w_proj.shape[0]
    p = max(k - stride, 0)
    Ho, Wo = (H + p - k) // stride + 1, (W + p - k) // stride + 1
    y = np.empty((B, Cout, Ho, Wo), np.float32)
    _lib.check(_lib.lib().cf_op_mbconv(device, _DT[dtype], ptr(x), ptr(w_exp), ptr(w_dw), ptr(w_proj), ptr(y),
                                       B, Cin, hid, Cout, H, W, k, stride), op=True)
    return y


def expand_dw(x, w_exp, w_dw, k, stride, dtype="bf16", device=0):
    """Expand 1x1 + Swish -> depthwise k x k + Swish of an MBConv block (model/centernet.py:109-114) as the
    single kernel the engine uses for the wide late blocks (bf16 storage only)."""
    x = f32(x)
    B, Cin, H, W = x.shape
    w_exp = f32(np.asarray(w_exp).reshape(w_exp.shape[0], -1))
    hid = w_exp.shape[0]
    w_dw = f32(np.asarray(w_dw).reshape(hid, k * k))
    p = max(k - stride, 0)
    Ho, Wo = (H + p - k) // stride + 1, (W + p - k) // stride + 1
    y = np.empty((B, hid, Ho, Wo), np.float32)
    _lib.check(_lib.lib().cf_op_expand_dw(device, _DT[dtype], ptr(x), ptr(w_exp), ptr(w_dw), ptr(y),
                                          B, Cin, hid, H, W, k, stride), op=True)
    return y


def stem(x, w, dtype="fp32", device=0):
    """first_conv (model/centernet.py:224): x uint8 [B,H,W,3] BGR (normalisation fused) or float32 [B,3,H,W]."""
    x = np.ascontiguousarray(x)
    if x.dtype == np.uint8:
        B, H, W, _ = x.shape
        fmt = _lib.CF_IN_U8_HWC_BGR
    else:
        x = f32(x)
        B, _, H, W = x.shape
        fmt = _lib.CF_IN_F32_NCHW
    y = np.empty((B, 32, H // 2, W // 2), np.float32)
    _lib.check(_lib.lib().cf_op_stem(device, _DT[dtype], ptr(x), fmt, ptr(f32(w)), ptr(y), B, H, W), op=True)
    return y


def _bn4(sd, prefix):
    return f32(np.stack([sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"],
                         sd[prefix + ".running_var"]]))


def idaup(lo, skip, sd, prefix, eps=1e-3, dtype="fp32", device=0):
    """IDAUp.forward (model/centernet.py:200-204) from a state_dict slice with raw BN parameters."""
    lo, skip = f32(lo), f32(skip)
    B, Cc, h, w = lo.shape
    Cs = skip.shape[1]
    y = np.empty((B, Cc, 2 * h, 2 * w), np.float32)
    w_up = f32(sd[prefix + ".up.weight"])
    w_cv = f32(np.asarray(sd[prefix + ".conv.0.weight"]).reshape(Cc, Cs))
    _lib.check(_lib.lib().cf_op_idaup(device, _DT[dtype], ptr(lo), ptr(skip), ptr(w_up), ptr(_bn4(sd, prefix + ".bn_up")),
                                      ptr(w_cv), ptr(_bn4(sd, prefix + ".conv.1")), float(eps), ptr(y),
                                      B, Cc, Cs, h, w), op=True)
    return y


def heads(x, sd, collapse=False, dtype="fp32", device=0):
    """The four heads (model/centernet.py:247-261) -> dict hm (raw), wh, lm, reg (NCHW)."""
    x = f32(x)
    B, _, h, w = x.shape
    names = ("hm", "wh", "lm", "reg")
    w0 = f32(np.stack([sd[n + ".0.weight"] for n in names]))
    b0 = f32(np.stack([sd[n + ".0.bias"] for n in names]))
    w1 = f32(np.concatenate([np.asarray(sd[n + ".1.weight"]).reshape(-1, 24) for n in names]))
    b1 = f32(np.concatenate([np.asarray(sd[n + ".1.bias"]) for n in names]))
    out = np.empty((B, 15, h, w), np.float32)
    _lib.check(_lib.lib().cf_op_heads(device, _DT[dtype], ptr(x), ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(out),
                                      B, h, w, 1 if collapse else 0), op=True)
    return {"hm": out[:, 0:1], "wh": out[:, 1:3], "lm": out[:, 3:13], "reg": out[:, 13:15]}


def shuffle_v2_block(x, sd, inp, oup, mid, ksize, stride, prefix="", dtype="fp32", device=0):
    """ShuffleV2Block(inp, oup, mid, ksize=, stride=).forward in eval mode (model/blocks.py:4-62) from a
    state_dict slice with the reference's key names (branch_main.0/1/3/4/5/6, branch_proj.0/1/2/3): ONE C-ABI call;
    BN fold, channel shuffle and concat all happen behind it."""
    x = f32(x)
    B, _, H, W = x.shape
    pad = ksize // 2
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    y = np.empty((B, oup, Ho, Wo), np.float32)

    def w(key):
        return f32(np.asarray(sd[prefix + key + ".weight"]))
    args = [w("branch_main.0"), _bn4(sd, prefix + "branch_main.1"), w("branch_main.3"), _bn4(sd, prefix + "branch_main.4"),
            w("branch_main.5"), _bn4(sd, prefix + "branch_main.6")]
    if stride == 2:
        args += [w("branch_proj.0"), _bn4(sd, prefix + "branch_proj.1"), w("branch_proj.2"), _bn4(sd, prefix + "branch_proj.3")]
    else:
        args += [None, None, None, None]
    _lib.check(_lib.lib().cf_op_shufflev2(device, _DT[dtype], ptr(x), ptr(y), B, int(inp), int(oup), int(mid), H, W,
                                          int(ksize), int(stride), *[ptr(a) for a in args]), op=True)
    return y


def ctdet_decode(heat, wh, reg=None, K=100, lm=None, device=0):
    """ctdet_decode (centerface_ext.py:52-82): (dets [B,K,6], lms [B,K,10]|None, inds [B,K] int64)."""
    heat, wh, reg, lm = f32(heat), f32(wh), f32(reg), f32(lm)
    B, _, h, w = heat.shape
    dets = np.empty((B, K, 6), np.float32)
    lms = np.empty((B, K, 10), np.float32) if lm is not None else None
    inds = np.empty((B, K), np.int64)
    _lib.check(_lib.lib().cf_op_ctdet_decode(device, ptr(heat), ptr(wh), ptr(reg), ptr(lm), B, h, w, int(K),
                                             ptr(dets), ptr(lms), ptr(inds)), op=True)
    return dets, lms, inds

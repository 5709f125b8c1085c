#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE in this container.

Run once, here (the reference is not present on the GPU box and never travels):

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens.py

What is written is data only: seeded inputs and the reference's outputs for them (plus the small
per-op weights that produced them).  The whole-network cases use
``centerface_amd.weights.synthetic_state_dict(0)`` loaded strictly into the reference's
``efficientnet_b0()``; only its fingerprint is stored, the weights are a pure function of the seed.

Stubs: mlconfig.register (decorator, model/centernet.py:178,298),
torchsummary.summary (:303), empty cv2 / torchvision / numba, and
model.centernet.ghost_net so that centerface_ext.py:4 resolves.
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def install_stubs():
    for name in ("mlconfig", "torchsummary", "cv2", "torchvision", "torchvision.transforms", "numba"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["mlconfig"].register = lambda f: f
    sys.modules["torchsummary"].summary = lambda *a, **k: None
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["numba"].jit = lambda *a, **k: (lambda f: f)
    # np.bool exists again in numpy >= 2.0 (alias of np.bool_), so centerface.py:119 needs no stub here
    sys.path.insert(0, REF)


def rnd(rng, *shape, scale=1.0):
    return (scale * rng.standard_normal(shape)).astype(np.float32)


def randomize_bn(mod, rng):
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            c = m.num_features
            m.running_mean.copy_(torch.from_numpy(rnd(rng, c, scale=0.1)))
            m.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
            m.weight.data.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
            m.bias.data.copy_(torch.from_numpy(rnd(rng, c, scale=0.1)))


def randomize_convs(mod, rng, gain=1.4):
    for m in mod.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            fan_in = m.weight[0].numel()
            m.weight.data.copy_(torch.from_numpy(rnd(rng, *m.weight.shape, scale=gain / np.sqrt(fan_in))))
            if m.bias is not None:
                m.bias.data.copy_(torch.from_numpy(rnd(rng, *m.bias.shape, scale=0.2)))


def sd_np(mod, prefix=""):
    return {prefix + k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}


@torch.no_grad()
def main():
    install_stubs()
    sys.path.insert(0, REPO)
    import model.centernet as cn
    cn.ghost_net = None
    import centerface_ext as ext
    import centerface as cf
    import model.blocks as blocks
    import centerface_amd as cfa

    os.makedirs(OUT, exist_ok=True)

    # ------------------------------------------------------------------ G3: whole network
    sd = cfa.weights.synthetic_state_dict(0)
    net = cn.efficientnet_b0().eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    fp = cfa.weights.fingerprint(sd)
    rng = np.random.default_rng(1234)
    g3 = {"weights_fingerprint": np.array(fp)}
    for tag, shape in (("a", (1, 3, 32, 32)), ("b", (2, 3, 64, 96)), ("c", (1, 3, 96, 64))):
        x = rnd(rng, *shape)
        out = net(torch.from_numpy(x))[0]
        g3["x_" + tag] = x
        for h in ("hm", "wh", "lm", "reg"):
            g3["%s_%s" % (h, tag)] = out[h].numpy().copy()
    # one uint8 image through the reference's own preprocessing arithmetic (centerface.py:32-37,
    # identity-resize case) -> network -> sigmoid/clamp (:43)
    img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    xi = (img.astype(np.float32) / 255.)
    xi = (xi - cf.CenterFace.mean) / cf.CenterFace.std
    xi = torch.unsqueeze(torch.FloatTensor(xi.transpose(2, 0, 1)), 0)
    out = net(xi)[0]
    g3["img_u8"] = img
    g3["img_hm_sigmoid"] = torch.clamp(out["hm"].sigmoid_(), min=1e-4, max=1 - 1e-4).numpy().copy()
    for h in ("wh", "lm", "reg"):
        g3["img_" + h] = out[h].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "net.npz"), **g3)

    # ------------------------------------------------------------------ G1/G2: ops and blocks
    rng = np.random.default_rng(77)
    ops = {}
    # MBConv configs: every distinct (t, k, s, residual) combination of the network at small size
    mb_cfgs = [  # cin, cout, t, k, s, H, W
        (32, 16, 1, 3, 1, 10, 12), (16, 24, 6, 3, 2, 12, 14), (24, 24, 6, 3, 1, 9, 11),
        (24, 32, 6, 5, 2, 12, 10), (32, 32, 6, 5, 1, 8, 9), (32, 64, 6, 3, 2, 7, 9),
        (64, 96, 6, 5, 1, 5, 6), (96, 160, 6, 5, 2, 6, 8), (160, 160, 6, 5, 1, 4, 5),
    ]
    names = []
    for i, (cin, cout, t, k, s, H, W) in enumerate(mb_cfgs):
        m = cn.MBConvBlock(cin, cout, expand_ratio=t, kernel_size=k, stride=s, se=False).eval()
        randomize_convs(m, rng)
        x = rnd(rng, 2, cin, H, W)
        tag = "mb%d" % i
        names.append(tag)
        ops[tag + "_cfg"] = np.array([cin, cout, t, k, s], np.int32)
        ops[tag + "_x"] = x
        ops[tag + "_y"] = m(torch.from_numpy(x)).numpy().copy()
        for kk, v in sd_np(m).items():
            ops[tag + "_w_" + kk] = v
    # ConvReLU (pad+conv+Swish): stem and each depthwise flavour incl. borders, non-square
    cr_cfgs = [(3, 32, 3, 2, 1, 10, 14), (16, 16, 3, 1, 16, 6, 7), (16, 16, 3, 2, 16, 7, 6),
               (24, 24, 5, 1, 24, 6, 9), (24, 24, 5, 2, 24, 9, 8), (8, 24, 1, 1, 1, 5, 4)]
    for i, (cin, cout, k, s, g, H, W) in enumerate(cr_cfgs):
        m = cn.ConvReLU(cin, cout, k, stride=s, groups=g).eval()
        randomize_convs(m, rng)
        x = rnd(rng, 2, cin, H, W)
        tag = "cr%d" % i
        ops[tag + "_cfg"] = np.array([cin, cout, k, s, g], np.int32)
        ops[tag + "_x"] = x
        ops[tag + "_y"] = m(torch.from_numpy(x)).numpy().copy()
        ops[tag + "_w"] = m[1].weight.detach().numpy().copy()
    # conv_1x1_bn
    m = cn.conv_1x1_bn(320, 24).eval()
    randomize_convs(m, rng); randomize_bn(m, rng)
    x = rnd(rng, 2, 320, 3, 4)
    ops["c1bn_x"] = x
    ops["c1bn_y"] = m(torch.from_numpy(x)).numpy().copy()
    for kk, v in sd_np(m, "conv_last.").items():
        ops["c1bn_w_" + kk] = v
    # IDAUp (random deconv weights, not the fill_up_weights init, so all four taps are exercised)
    for i, ch in enumerate((96, 32, 24)):
        m = cn.IDAUp(24, ch).eval()
        randomize_convs(m, rng); randomize_bn(m, rng)
        lo, sk = rnd(rng, 2, 24, 3, 5), rnd(rng, 2, ch, 6, 10)
        tag = "ida%d" % i
        ops[tag + "_lo"], ops[tag + "_skip"] = lo, sk
        ops[tag + "_y"] = m(torch.from_numpy(lo), torch.from_numpy(sk)).numpy().copy()
        for kk, v in sd_np(m, "up.").items():
            ops[tag + "_w_" + kk] = v
    # fill_up_weights result (model/centernet.py:168-177) on a 2x2 depthwise deconv
    up = torch.nn.ConvTranspose2d(4, 4, 2, 2, 0, 0, 4, bias=False)
    cn.fill_up_weights(up)
    ops["fill_up_2x2"] = up.weight.detach().numpy().copy()
    # one head pair conv3x3+b -> conv1x1+b, via a full EfficientNet's lm head module
    head = net.lm
    x = rnd(rng, 2, 24, 7, 9)
    ops["head_x"] = x
    ops["head_y"] = head(torch.from_numpy(x)).numpy().copy()
    # ShuffleV2Block s1 / s2, k3 / k5 (model/blocks.py:4-62), eval mode with randomised BN
    for i, (inp, oup, mid, k, s, H, W) in enumerate(((24, 48, 24, 3, 1, 6, 7), (24, 48, 24, 5, 2, 8, 6),
                                                     (16, 32, 16, 5, 1, 5, 5), (16, 40, 24, 3, 2, 7, 9))):
        m = blocks.ShuffleV2Block(inp if s == 2 else oup // 2, oup, mid, ksize=k, stride=s).eval()
        randomize_convs(m, rng); randomize_bn(m, rng)
        cin = inp if s == 2 else oup
        x = rnd(rng, 2, cin, H, W)
        tag = "sh%d" % i
        ops[tag + "_cfg"] = np.array([inp if s == 2 else oup // 2, oup, mid, k, s], np.int32)
        ops[tag + "_x"] = x
        ops[tag + "_y"] = m(torch.from_numpy(x)).numpy().copy()
        for kk, v in sd_np(m).items():
            ops[tag + "_w_" + kk] = v
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **ops)

    # ------------------------------------------------------------------ G4: decoder D3
    rng = np.random.default_rng(99)
    d3 = {}
    for tag, (B, H, W, K) in (("s", (2, 16, 24, 10)), ("m", (3, 40, 56, 100)), ("l", (2, 160, 160, 100)),
                              ("x", (1, 96, 128, 1000))):
        # strictly distinct scores: a random permutation of an evenly spaced grid in (1e-4, 1)
        n = B * H * W
        vals = np.linspace(2e-4, 0.999, n, dtype=np.float64).astype(np.float32)
        assert len(np.unique(vals)) == n
        heat = vals[rng.permutation(n)].reshape(B, 1, H, W)
        wh = np.abs(rnd(rng, B, 2, H, W, scale=3.0)) + 1
        reg = rng.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
        det = ext.ctdet_decode(torch.from_numpy(heat), torch.from_numpy(wh), torch.from_numpy(reg), K=K)
        det_noreg = ext.ctdet_decode(torch.from_numpy(heat), torch.from_numpy(wh), None, K=K)
        tk = ext._topk(ext._nms(torch.from_numpy(heat)), K=K)
        d3[tag + "_heat"], d3[tag + "_wh"], d3[tag + "_reg"] = heat, wh, reg
        d3[tag + "_K"] = np.array(K)
        d3[tag + "_det"] = det.numpy().copy()
        d3[tag + "_det_noreg"] = det_noreg.numpy().copy()
        d3[tag + "_nms"] = ext._nms(torch.from_numpy(heat)).numpy().copy()
        for nm, t in zip(("score", "inds", "clses", "ys", "xs"), tk):
            d3["%s_topk_%s" % (tag, nm)] = t.numpy().copy()
    # plateau case: _nms keeps every cell of a plateau (flagged: order among equals is unspecified)
    heat = np.full((1, 1, 6, 6), 1e-4, np.float32)
    heat[0, 0, 2, 2] = heat[0, 0, 2, 3] = 0.7
    heat[0, 0, 4, 5] = 0.9
    d3["tie_heat"] = heat
    d3["tie_nms"] = ext._nms(torch.from_numpy(heat)).numpy().copy()
    np.savez_compressed(os.path.join(OUT, "decode_d3.npz"), **d3)

    # ------------------------------------------------------------------ G5/G6: decoder D1, nms, transform
    rng = np.random.default_rng(5)
    d1 = {}
    face = object.__new__(cf.CenterFace)
    face.landmarks = True
    # (a) the survey's probe sample (Appendix C): border clamps exercised
    H = W = 160
    hm = np.full((1, 1, H, W), 1e-4, np.float32)
    for (y, x), v in (((10, 20), 0.9), ((10, 21), 0.8), ((80, 80), 0.7), ((159, 159), 0.6), ((0, 0), 0.5)):
        hm[0, 0, y, x] = v
    wh = rng.uniform(2, 22, (1, 2, H, W)).astype(np.float32)
    off = rng.uniform(0, 1, (1, 2, H, W)).astype(np.float32)
    lm = rnd(rng, 1, 10, H, W, scale=2.0)
    b, l = face.decode(hm, wh, off, lm, (640, 640), threshold=0.05)
    d1["a_hm"], d1["a_wh"], d1["a_off"], d1["a_lm"] = hm, wh, off, lm
    d1["a_size"] = np.array([640, 640])
    d1["a_boxes"], d1["a_lms"] = np.asarray(b, np.float32), np.asarray(l, np.float32)
    # (b) dense random case, non-square, many overlaps -> NMS does real work
    H, W = 24, 40
    hm = rng.uniform(1e-4, 0.999, (1, 1, H, W)).astype(np.float32)
    wh = rng.uniform(0.5, 6, (1, 2, H, W)).astype(np.float32)
    off = rng.uniform(0, 1, (1, 2, H, W)).astype(np.float32)
    lm = rnd(rng, 1, 10, H, W, scale=2.0)
    b, l = face.decode(hm, wh, off, lm, (96, 160), threshold=0.9)   # threshold is ignored by the reference
    d1["b_hm"], d1["b_wh"], d1["b_off"], d1["b_lm"] = hm, wh, off, lm
    d1["b_size"] = np.array([96, 160])
    d1["b_boxes"], d1["b_lms"] = np.asarray(b, np.float32), np.asarray(l, np.float32)
    # (c) empty case: returns two python lists
    hm = np.full((1, 1, 8, 8), 0.2, np.float32)
    b, l = face.decode(hm, np.ones((1, 2, 8, 8), np.float32), np.zeros((1, 2, 8, 8), np.float32),
                       np.zeros((1, 10, 8, 8), np.float32), (32, 32))
    d1["c_empty_is_list"] = np.array([isinstance(b, list) and len(b) == 0, isinstance(l, list) and len(l) == 0])
    # (d) nms alone
    n = 300
    xy = rng.uniform(0, 100, (n, 2)).astype(np.float32)
    sz = rng.uniform(5, 40, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + sz], 1).astype(np.float32)
    scores = rng.permutation(np.linspace(0.05, 0.99, n)).astype(np.float32)
    d1["nms_boxes"], d1["nms_scores"] = boxes, scores
    for thr in (0.3, 0.5):
        d1["nms_keep_%d" % int(thr * 10)] = np.asarray(face.nms(boxes, scores, thr), np.int64)
    # (e) transform table (centerface.py:68-71)
    sizes = [(478, 720), (640, 640), (1000, 750), (353, 490), (898, 1600), (32, 32), (33, 31), (1280, 1280)]
    d1["tf_in"] = np.asarray(sizes, np.int64)
    d1["tf_out"] = np.asarray([face.transform(h, w) for h, w in sizes], np.float64)
    # (f) the floor-division rescale of __call__ (centerface.py:55-58) on decoded boxes of case (a)
    dets, lms_ = d1["a_boxes"].copy(), d1["a_lms"].copy()
    sw, sh = np.float64(736 / 720), np.float64(480 / 478)
    dets[:, 0:4:2], dets[:, 1:4:2] = dets[:, 0:4:2] // sw, dets[:, 1:4:2] // sh
    lms_[:, 0:10:2], lms_[:, 1:10:2] = lms_[:, 0:10:2] // sw, lms_[:, 1:10:2] // sh
    d1["f_scale"] = np.array([sh, sw])
    d1["f_dets"], d1["f_lms"] = dets, lms_
    np.savez_compressed(os.path.join(OUT, "decode_d1.npz"), **d1)

    # ------------------------------------------------------------------ decoder D2 (eval_widerface.py:92-152)
    import eval_widerface as ew
    rng = np.random.default_rng(17)
    d2 = {}
    for tag, (H, W, thr) in (("a", (24, 40, 0.6)), ("b", (160, 160, 0.97)), ("c", (20, 12, 0.05))):
        # strictly distinct scores: np.argsort(scores)[::-1] (an unstable sort) leaves the order of equal
        # scores implementation-defined, and the greedy NMS result depends on it
        vals = np.linspace(2e-4, 0.999, H * W, dtype=np.float64).astype(np.float32)
        assert len(np.unique(vals)) == H * W
        hm = vals[rng.permutation(H * W)].reshape(1, H, W)
        wh = rng.uniform(0.5, 8, (2, H, W)).astype(np.float32)
        off = rng.uniform(-0.5, 1.0, (2, H, W)).astype(np.float32)
        boxes = ew.decode(hm, wh, off, None, (H * 4, W * 4), threshold=thr)
        d2[tag + "_hm"], d2[tag + "_wh"], d2[tag + "_off"] = hm, wh, off
        d2[tag + "_thr"] = np.array(thr)
        d2[tag + "_boxes"] = np.asarray(boxes, np.float32)
    d2["empty_is_list"] = np.array(isinstance(ew.decode(np.full((1, 4, 4), 0.1, np.float32), np.ones((2, 4, 4), np.float32),
                                                         np.zeros((2, 4, 4), np.float32), None, (16, 16), threshold=0.5), list))
    np.savez_compressed(os.path.join(OUT, "decode_d2.npz"), **d2)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
    leftovers = [r for r, d, _ in os.walk(REF) if "__pycache__" in d]
    assert not leftovers, "bytecode written into the reference tree: %s" % leftovers


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Pin oracle/bf16_emulation.py to the REFERENCE: run the reference's own module graph
(model.centernet.efficientnet_b0, model/centernet.py:205-280) with

  * weights quantised exactly as the bf16 engine's packers quantise them (expand/stem weights as
    bf16(-log2(e) w) / -log2(e), depthwise taps as fp16, project weights as bf16(-ln2 w) / -ln2 or bf16(w),
    BN folded into the 1x1 convs in float64 then bf16, heads collapsed in float64 then bf16 and written back
    into the reference's conv3x3 -> conv1x1 pair as [W_collapsed ; 0] -> [I | 0]), and
  * forward hooks on the reference's modules (Swish outputs of MBConvBlock.conv[0] / conv[1], block
    outputs, conv_last, IDAUp outputs -- model/centernet.py:89-140,179-204,263-280) that round the
    activations at the engine's storage points (fp16 round-toward-zero for the expanded tile, bf16 elsewhere),

and store inputs + outputs in tests/golden/net_bf16emu.npz.  The hooks work at TRUE scale (they scale by
-log2(e), round, and scale back), the oracle works in the engine's pre-scaled arithmetic: two independent
formulations; they agree up to 1-ulp rounding flips caused by fp32 noise.

Run here (the reference never travels):   PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens_bf16emu.py
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, REPO)
from gen_goldens import install_stubs, OUT          # noqa: E402


@torch.no_grad()
def main():
    install_stubs()
    import model.centernet as cn
    cn.ghost_net = None
    import centerface_amd as cfa
    from oracle import bf16_emulation as E

    C1, C2 = float(E.NEG_LOG2E), float(E.NEG_LN2)
    f32 = torch.float32

    def qscale(w, c):                   # bf16(c * w) / c, the value the reference must multiply by at true scale
        return E.q_bf16(torch.as_tensor(w, dtype=f32) * torch.tensor(c, dtype=f32)) / torch.tensor(c, dtype=f32)

    sd = cfa.weights.synthetic_state_dict(0)
    tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    net = cn.efficientnet_b0().eval()
    net.load_state_dict(tsd, strict=True)

    split = set(E.SPLIT_BLOCKS)
    hooks = []

    def round_scaled_f16(mod, inp, out):     # expanded tile: fp16 RTZ of -log2(e) * swish
        return E.q_f16_rtz_sat(out * C1) / C1

    def round_scaled_bf16(mod, inp, out):    # fused blocks' project operand: bf16 of -log2(e) * swish(dw)
        return E.q_bf16(out * C1) / C1

    def round_bf16(mod, inp, out):
        return E.q_bf16(out)

    def identity_bn(bn, shift):
        bn.weight.fill_(1.0); bn.bias.copy_(shift)
        bn.running_mean.zero_(); bn.running_var.fill_(1.0 - bn.eps)

    # ---- stem + MBConv blocks
    net.first_conv[0][1].weight.copy_(qscale(net.first_conv[0][1].weight, C1))
    hooks.append(net.first_conv[0][2].register_forward_hook(round_scaled_f16))          # stem Swish -> tile
    for li in range(7):
        for i, blk in enumerate(getattr(net, "layer%d" % li)):
            name = "layer%d.%d" % (li, i)
            seq = blk.conv
            if li == 0:
                dw, proj = seq[0], seq[1]
            else:
                exp, dw, proj = seq[0], seq[1], seq[2]
                exp[1].weight.copy_(qscale(exp[1].weight, C1))
                hooks.append(exp[2].register_forward_hook(round_scaled_f16))
            dw[1].weight.copy_(E.q_f16_rne_sat(dw[1].weight))
            if name in split:
                hooks.append(dw[2].register_forward_hook(round_bf16))                   # depthwise output in HBM
                proj.weight.copy_(E.q_bf16(proj.weight))
            else:
                hooks.append(dw[2].register_forward_hook(round_scaled_bf16))
                proj.weight.copy_(qscale(proj.weight, C2))
            hooks.append(blk.register_forward_hook(round_bf16))                         # block output (after the residual)

    # ---- conv_last / IDAUp: BN folded in float64 (cf_runtime.hip bn_fold), BN modules made the identity + shift
    wq, shift = E.folded_pw(tsd, "conv_last.0.weight", "conv_last.1", net.conv_last[1].eps)
    net.conv_last[0].weight.copy_(wq); identity_bn(net.conv_last[1], shift)
    hooks.append(net.conv_last.register_forward_hook(round_bf16))
    for n in ("up1", "up2", "up3"):
        up = getattr(net, n)
        wq, shift = E.folded_pw(tsd, n + ".conv.0.weight", n + ".conv.1", up.conv[1].eps)
        up.conv[0].weight.copy_(wq); identity_bn(up.conv[1], shift)
        scale, shift_up = E.bn_fold(tsd, n + ".bn_up", up.bn_up.eps)
        up.up.weight.copy_((tsd[n + ".up.weight"].double() * scale.reshape(-1, 1, 1, 1)).float())
        identity_bn(up.bn_up, shift_up.float())
        hooks.append(up.register_forward_hook(round_bf16))

    # ---- heads: the collapsed 3x3 24->15 conv expressed in the reference's conv3x3 -> conv1x1 pair
    W, b = E.collapsed_head_weights(tsd)
    o = 0
    for hn, c in (("hm", 1), ("wh", 2), ("lm", 10), ("reg", 2)):
        fc = getattr(net, hn)
        fc[0].weight.zero_(); fc[0].bias.zero_(); fc[1].weight.zero_(); fc[1].bias.zero_()
        fc[0].weight[:c].copy_(W[o:o + c]); fc[0].bias[:c].copy_(b[o:o + c])
        for j in range(c):
            fc[1].weight[j, j, 0, 0] = 1.0
        o += c

    # every block output (bf16-valued -> stored as raw bf16 bits) so the CPU test can check the oracle block by
    # block on the hooked reference's OWN inputs ("teacher forcing": end to end the two formulations drift apart
    # through 1-ulp rounding flips -- a quantised network is chaotic at that level -- block by block they do not)
    feats = {}

    def grab(name):
        return lambda m, i, out: feats.__setitem__(name, out.numpy().copy())
    for li in range(7):
        for i, blk in enumerate(getattr(net, "layer%d" % li)):
            hooks.append(blk.register_forward_hook(grab("layer%d.%d" % (li, i))))
    for n in ("conv_last", "up1", "up2", "up3"):
        hooks.append(getattr(net, n).register_forward_hook(grab(n)))

    def bf16_bits(a):
        assert np.array_equal(E.q_bf16(torch.from_numpy(a)).numpy(), a)
        return (a.view(np.uint32) >> 16).astype(np.uint16)

    rng = np.random.default_rng(4321)
    g = {"weights_fingerprint": np.array(cfa.weights.fingerprint(sd))}
    for tag, shape in (("a", (1, 3, 32, 32)), ("b", (1, 3, 64, 96)), ("c", (1, 3, 160, 96))):
        x = (rng.standard_normal(shape)).astype(np.float32)
        out = net(E.q_bf16(torch.from_numpy(x)))[0]                                    # input quantisation = the stem's staging
        g["x_" + tag] = x
        for h in ("hm", "wh", "lm", "reg"):
            g["%s_%s" % (h, tag)] = out[h].numpy().copy()
        for n, v in feats.items():
            g["%s_%s" % (n, tag)] = bf16_bits(v)
    # uint8 image through the engine's one-fma normalisation
    img = rng.integers(0, 256, (1, 96, 64, 3), dtype=np.uint8)
    out = net(E.normalise_u8(img))[0]
    g["img_u8"] = img
    for h in ("hm", "wh", "lm", "reg"):
        g["img_%s" % h] = out[h].numpy().copy()
    for h in hooks:
        h.remove()
    path = os.path.join(OUT, "net_bf16emu.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes,", len(g), "arrays")

    # self-check against the oracle formulation: end to end, then block by block on the golden's own inputs
    for tag in ("a", "b", "c"):
        o2, f2 = E.forward(sd, x=g["x_" + tag], return_features=True)
        for h in ("hm", "wh", "lm", "reg"):
            d = np.abs(o2[h].numpy() - g["%s_%s" % (h, tag)])
            print(tag, h, "end-to-end max|d| %.3e mean %.3e  (rms %.3f)" % (d.max(), d.mean(), np.sqrt((g["%s_%s" % (h, tag)] ** 2).mean())))
        worst = E.check_blockwise(sd, {k[:-2]: v for k, v in g.items() if k.endswith("_" + tag)})
        print(tag, "block by block, worst |d| / tolerance:", {k: round(v, 3) for k, v in worst.items()})


if __name__ == "__main__":
    main()

"""Checkpoint schema of the CenterFace network (94 tensors) and the block table it is derived from.

The reference builds the network in ``model/centernet.py:205-261`` (EfficientNet-B0 settings table
``:211-219``, SE disabled ``:232``, no BN inside MBConv ``:117-120``) and saves a plain
``state_dict`` (``train.py:165``) that ``centerface.py:23-24`` loads strictly.  This module
restates only the *names and shapes* so the loader can validate a user-supplied checkpoint and
the synthetic-weight generator can produce a schema-identical one.  The graph itself is executed
by the C++/HIP runtime (``csrc/cf_runtime.hip``), which carries the same table.
"""
from collections import OrderedDict

# (expand t, out channels c, repeats n, first stride s, dw kernel k) -- model/centernet.py:211-219
MBCONV_SETTINGS = (
    (1, 16, 1, 1, 3),
    (6, 24, 2, 2, 3),
    (6, 32, 2, 2, 5),
    (6, 64, 2, 2, 3),
    (6, 96, 2, 1, 5),
    (6, 160, 2, 2, 5),
    (6, 320, 1, 1, 3),
)
STEM_OUT = 32          # model/centernet.py:223-224
NECK = 24              # conv_last / IDAUp width, model/centernet.py:236-239
IDA_SKIPS = (("up1", 96), ("up2", 32), ("up3", 24))   # model/centernet.py:237-239
HEADS = OrderedDict((("hm", 1), ("wh", 2), ("lm", 10), ("reg", 2)))  # model/centernet.py:240-245
BN_EPS_CONV_LAST = 1e-5   # nn.BatchNorm2d default, model/centernet.py:182
BN_EPS_IDA = 1e-3         # model/centernet.py:193,197
DOWN_RATIO = 4            # heads live on the stride-4 map (centerface.py:84,88)


def mbconv_blocks():
    """Yield (prefix, cin, cout, t, k, s) for the 12 MBConv blocks in execution order."""
    cin = STEM_OUT
    for li, (t, c, n, s, k) in enumerate(MBCONV_SETTINGS):
        for i in range(n):
            yield ("layer%d.%d" % (li, i), cin, c, t, k, s if i == 0 else 1)
            cin = c


def _bn(prefix, c, out):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    out[prefix + ".running_mean"] = (c,)
    out[prefix + ".running_var"] = (c,)
    out[prefix + ".num_batches_tracked"] = ()


def state_dict_schema():
    """OrderedDict name -> shape, in the order ``efficientnet_b0().state_dict()`` yields them."""
    sd = OrderedDict()
    sd["first_conv.0.1.weight"] = (STEM_OUT, 3, 3, 3)
    for prefix, cin, cout, t, k, _s in mbconv_blocks():
        hid = cin * t
        j = 0
        if t != 1:
            sd["%s.conv.0.1.weight" % prefix] = (hid, cin, 1, 1)
            j = 1
        sd["%s.conv.%d.1.weight" % (prefix, j)] = (hid, 1, k, k)
        sd["%s.conv.%d.weight" % (prefix, j + 1)] = (cout, hid, 1, 1)
    sd["conv_last.0.weight"] = (NECK, MBCONV_SETTINGS[-1][1], 1, 1)
    _bn("conv_last.1", NECK, sd)
    for name, skip in IDA_SKIPS:
        sd[name + ".up.weight"] = (NECK, 1, 2, 2)
        _bn(name + ".bn_up", NECK, sd)
        sd[name + ".conv.0.weight"] = (NECK, skip, 1, 1)
        _bn(name + ".conv.1", NECK, sd)
    for head, c in HEADS.items():
        sd[head + ".0.weight"] = (NECK, NECK, 3, 3)
        sd[head + ".0.bias"] = (NECK,)
        sd[head + ".1.weight"] = (c, NECK, 1, 1)
        sd[head + ".1.bias"] = (c,)
    return sd

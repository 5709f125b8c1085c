"""Weights for the CenterFace runtime: schema validation, checkpoint loading, synthetic generator.

The reference loads ``weight/model_epoch_100.pt`` (``centerface.py:23-24``, strict); that file is
not distributed with it (``.MISSING_LARGE_BLOBS``).  A user who has one can pass its path; for
tests and benchmarks ``synthetic_state_dict(seed)`` produces a schema-identical set of tensors
that is a pure function of the seed (numpy ``default_rng``), scaled so that activations stay O(1)
through all 16 blocks -- default-initialised weights make every deep-layer bug invisible because
the signal collapses.

All tensors are numpy float32 (``num_batches_tracked`` int64), layout exactly as PyTorch stores
them (conv weights OIHW).  BN folding and repacking to kernel layouts happen in the C++ runtime.
"""
import hashlib
from collections import OrderedDict

import numpy as np

from .schema import state_dict_schema, HEADS

# Second-moment gains found once with the oracle so that |activation| RMS stays within [0.3, 3]
# from first_conv to the heads (checked by tests/test_weights.py::test_activation_scale).
_GAIN_SWISH = 1.5       # conv followed by Swish (between the ReLU-like 1.41 and the small-signal 1.68)
_GAIN_LINEAR = 1.0
_GAIN_RESIDUAL = 0.6    # linear project convs on residual blocks (keeps x + f(x) from growing)


def _is_residual_project(name, schema):
    # a project conv "layerL.i.conv.J.weight" whose block has in==out and stride 1, i.e. i >= 1 here
    parts = name.split(".")
    return parts[0].startswith("layer") and parts[1] != "0" and len(parts) == 5 and parts[2] == "conv"


def synthetic_state_dict(seed=0):
    """Deterministic, variance-calibrated synthetic weights with the reference's 94-tensor schema."""
    schema = state_dict_schema()
    sd = OrderedDict()
    for idx, (name, shape) in enumerate(schema.items()):
        rng = np.random.default_rng([int(seed), idx])
        leaf = name.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[name] = np.array(100, dtype=np.int64)
        elif leaf == "running_mean":
            sd[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "running_var":
            sd[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1 and leaf == "weight":          # BN gamma
            sd[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1 and leaf == "bias":
            sd[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:                                               # conv / deconv weight, OIHW
            fan_in = int(np.prod(shape[1:]))
            top = name.split(".")[0]
            if name.endswith(".up.weight"):                 # depthwise 2x2 deconv: 1 tap per output
                std = 1.0
            elif top in HEADS:
                std = _GAIN_LINEAR / np.sqrt(fan_in)
            elif top.startswith("up") or top == "conv_last":
                std = 1.4 / np.sqrt(fan_in)
            elif ".conv." in name and len(name.split(".")) == 5:   # MBConv project (linear)
                g = _GAIN_RESIDUAL if _is_residual_project(name, schema) else _GAIN_LINEAR
                std = g / np.sqrt(fan_in)
            else:                                           # stem, expand, depthwise (+Swish)
                std = _GAIN_SWISH / np.sqrt(fan_in)
            sd[name] = (std * rng.standard_normal(shape)).astype(np.float32)
    # sparse but non-empty detection set (~1 % of cells above 0.3): push the heat-map logit
    # negative (the reference initialises the hm bias to -1.79, model/centernet.py:257-258)
    sd["hm.1.bias"] = np.full((1,), -4.0, dtype=np.float32)
    return sd


def validate_state_dict(sd):
    """Strict check (same spirit as load_state_dict(strict=True), centerface.py:24). Returns an
    OrderedDict of contiguous numpy arrays in schema order."""
    schema = state_dict_schema()
    missing = [k for k in schema if k not in sd]
    extra = [k for k in sd if k not in schema]
    if missing or extra:
        raise ValueError("state_dict mismatch: missing=%s unexpected=%s" % (missing[:5], extra[:5]))
    out = OrderedDict()
    for name, shape in schema.items():
        v = sd[name]
        if hasattr(v, "detach"):            # torch tensor
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if tuple(v.shape) != tuple(shape):
            raise ValueError("size mismatch for %s: got %s, expected %s" % (name, v.shape, shape))
        if name.endswith("num_batches_tracked"):
            out[name] = np.ascontiguousarray(v, dtype=np.int64)
        else:
            out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def load_checkpoint(path):
    """Load a reference-format checkpoint (``torch.save(model.state_dict())``, train.py:165)."""
    import torch  # plumbing only: deserialisation of the .pt container
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd and "hm.0.weight" not in sd:
        sd = sd["state_dict"]
    return validate_state_dict(sd)


def fingerprint(sd):
    """sha256 over names, shapes and raw bytes -- used to pin the synthetic recipe in goldens."""
    h = hashlib.sha256()
    for k, v in sd.items():
        a = np.ascontiguousarray(v)
        h.update(k.encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()

"""MI355X-native CenterFace inference hot path (drop-in for the reference's ``centerface.py``).

    from centerface_amd import CenterFace
    dets, lms = CenterFace(640, 640)(img_bgr_u8)
"""
from . import schema, weights, _lib, ops, distributed, post_process, eval_widerface, demo, losses  # noqa: F401
from .centerface import CenterFace, CenterFaceBuckets, Engine, EngineRing, pin, unpin, pinned_empty, pinned_copy, is_pinned  # noqa: F401

__all__ = ["CenterFace", "CenterFaceBuckets", "Engine", "EngineRing", "pin", "unpin", "pinned_empty", "pinned_copy", "is_pinned", "schema", "weights", "ops"]

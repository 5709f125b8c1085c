"""MI355X-native CenterFace inference hot path (drop-in for the reference's ``centerface.py``)."""
from . import schema, weights  # noqa: F401

__all__ = ["schema", "weights"]

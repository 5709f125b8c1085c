"""Import shim: exposes the package that lives in ``lightweight-face-detection-centernet_amd/``
(a directory name Python cannot import directly because of the hyphens) as ``centerface_amd``.

    import centerface_amd
    det = centerface_amd.CenterFace(640, 640)

Nothing else lives here; the product code is in the package directory.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "lightweight-face-detection-centernet_amd")
_NAME = "centerface_amd"

if not (_NAME in sys.modules and getattr(sys.modules[_NAME], "__path__", None)):
    _spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"),
        submodule_search_locations=[_PKG_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)

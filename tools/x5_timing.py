"""Phase timing of expdw_f32_kernel (cf_mbconv5.hip) on the production shapes, B = 64: needs a library built with -DCF_X5_TIMING
(tools/ab_build.sh x5t "-DCF_X5_TIMING" cf_mbconv5.hip cf_ops.hip; CF_LIB=ab/x5t/libcenterface_hip.so python tools/x5_timing.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import centerface_amd as cfa
from centerface_amd import ops
for name, cin, k, s, H in (("4.0", 64, 5, 1, 40), ("4.1", 96, 5, 1, 40), ("5.0", 96, 5, 2, 40), ("5.1", 160, 5, 1, 20), ("6.0", 160, 3, 1, 20)):
    rng = np.random.default_rng(1)
    hid = cin * 6
    we = (rng.standard_normal((hid, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    wd = (rng.standard_normal((hid, 1, k, k)) / k).astype(np.float32)
    x = rng.standard_normal((64, cin, H, H)).astype(np.float32)
    sys.stderr.write("layer%s: " % name); sys.stderr.flush()
    ops.expand_dw(x, we, wd, k, s, dtype="fp32_split")
# the fused blocks of the split mode (cf_mbconv4.hip stamps: same build flag)
for name, cin, cout, k, s, H in (("1.0", 16, 24, 3, 2, 320), ("1.1", 24, 24, 3, 1, 160), ("2.0", 24, 32, 5, 2, 160), ("2.1", 32, 32, 5, 1, 80)):
    rng = np.random.default_rng(2)
    hid = cin * 6
    we = (rng.standard_normal((hid, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
    wd = (rng.standard_normal((hid, 1, k, k)) / k).astype(np.float32)
    wp = (rng.standard_normal((cout, hid, 1, 1)) / np.sqrt(hid)).astype(np.float32)
    x = rng.standard_normal((64, cin, H, H)).astype(np.float32)
    sys.stderr.write("layer%s: " % name); sys.stderr.flush()
    ops.mbconv(x, we, wd, wp, k, s, dtype="fp32_split")
# the split / exact stem (stem0_kernel): stamps accumulated in a device symbol over one forward at B = 64
import ctypes
L = cfa._lib.lib()
if hasattr(L, "cf_debug_stem_stamps"):
    for dt in ("fp32_split", "fp32"):
        eng = cfa.Engine(640, 640, max_batch=64, dtype=dt)
        imgs = np.random.default_rng(0).integers(0, 256, (64, 640, 640, 3), dtype=np.uint8)
        eng.forward_enqueue(imgs); eng.synchronize()
        z = (ctypes.c_ulonglong * 6)()
        L.cf_debug_stem_stamps(z, 1)
        eng.forward_enqueue(imgs); eng.synchronize()
        L.cf_debug_stem_stamps(z, 1)
        n = max(1, z[5])
        sys.stderr.write("stem0_kernel<%s>: %d waves; mean cycles per wave: staging %.0f  barrier %.0f  stem conv + Swish %.0f  barrier %.0f  depthwise + project %.0f\n"
                         % (dt, n, z[0] / n, z[1] / n, z[2] / n, z[3] / n, z[4] / n))
        eng.close()

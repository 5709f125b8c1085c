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

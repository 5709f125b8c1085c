#!/usr/bin/env python3
"""Write a state_dict as the flat tensor file examples/detect.c reads (name[64], int32 dtype, int32 ndim,
int64 dims[4], raw data -- in state_dict order).

    python tools/export_weights.py out.bin [checkpoint.pt]      (default: the calibrated synthetic weights, seed 0)
"""
import os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa


def export(path, sd):
    with open(path, "wb") as f:
        for name, v in sd.items():
            v = np.ascontiguousarray(v)
            dt = 1 if v.dtype == np.int64 else 0
            if dt == 0:
                v = v.astype(np.float32)
            dims = list(v.shape) + [0] * (4 - v.ndim)
            f.write(name.encode().ljust(64, b"\0"))
            f.write(struct.pack("<ii4q", dt, v.ndim, *dims))
            f.write(v.tobytes())


if __name__ == "__main__":
    sd = cfa.weights.load_checkpoint(sys.argv[2]) if len(sys.argv) > 2 else cfa.weights.synthetic_state_dict(0)
    export(sys.argv[1], sd)
    print("wrote %s (%d tensors)" % (sys.argv[1], len(sd)))

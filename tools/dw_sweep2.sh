#!/bin/bash
# every tile of the strip kernel's list x LDS caps: GB/s per layer (experiments build); one line per (tile, cap)
export CF_LIB=$PWD/lightweight-face-detection-centernet_amd/libcenterface_hip_exp.so
for tile in 0 1 2 3 4 5 6 7 8; do for cap in 16 24 32 48 60; do
  echo -n "tile=$tile cap=$cap "
  CF_DW_TILE=$tile CF_DW_CAP=$cap python3 tools/profile_ops.py --no-fuse --reps 3 2>/dev/null | grep -E "\.dw" | awk '{printf "%s=%d ", substr($1,6,3), $4}'
  echo
done; done

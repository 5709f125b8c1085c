#!/bin/bash
# tools/dw_sweep.sh: standalone depthwise (unfused path) GB/s per layer over LDS caps / persistent modes (experiments build)
export CF_LIB=$PWD/lightweight-face-detection-centernet_amd/libcenterface_hip_exp.so
for cfg in "CF_DW_STRIP=0" "CF_DW_CAP=12" "CF_DW_CAP=16" "CF_DW_CAP=24" "CF_DW_CAP=32" "CF_DW_CAP=48" "CF_DW_CAP=60" "CF_DW_CAP=30 CF_DW_WGS=2" "CF_DW_CAP=18 CF_DW_WGS=4" "CF_DW_CAP=12 CF_DW_WGS=6"; do
  echo "== $cfg"
  env $cfg python3 tools/profile_ops.py --no-fuse 2>/dev/null | grep -E "\.dw" | awk '{printf "%s %s %s | ", $1, $4, $6" "$7" "$8" "$9" "$10}' | sed 's/void cf::dw_//g; s/_kernel<unsigned short,//g; s/>(cf::DwParams,//g'
  echo
done

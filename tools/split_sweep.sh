#!/bin/bash
# A/B sweep of the split-mode (fp32_split) block kernels on the GPU box: experiments build of the library, one profile per variant.
#   make -C lightweight-face-detection-centernet_amd/csrc EXP=1 ; tools/split_sweep.sh > gpurun_out/split_sweep.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export CF_LIB=$ROOT/lightweight-face-detection-centernet_amd/libcenterface_hip_exp.so
for v in 0 1 3 4 5 6; do
  echo "== CF_F4_VARIANT=$v"; CF_F4_VARIANT=$v python $ROOT/tools/profile_ops.py --dtype fp32_split --reps 3 | grep -E "mbconv|sum of"
done
for v in 1 2 3; do
  echo "== CF_MB_VARIANT=$v"; CF_MB_VARIANT=$v python $ROOT/tools/profile_ops.py --dtype fp32_split --reps 3 | grep -E "mbconv|sum of"
done

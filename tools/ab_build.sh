#!/bin/bash
# Build a VARIANT of libcenterface_hip.so next to the product build, for A/B timing on the GPU box:
#   tools/ab_build.sh <name> "<extra hipcc flags, e.g. -DCF_ABL=1>" [file.hip ...]   ->  ab/<name>/libcenterface_hip.so
#   CF_LIB=$PWD/ab/<name>/libcenterface_hip.so python tools/profile_ops.py
# Only the listed sources (default: cf_mbconv2.hip) are recompiled with the extra flags; the other objects are reused.
set -e
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-cf_mbconv2.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/lightweight-face-detection-centernet_amd/csrc
OUT=$ROOT/ab/$NAME
mkdir -p "$OUT"
make -C "$SRC" -j8 > /dev/null
OBJS=""
for o in "$SRC"/build/*.o; do
  b=$(basename "$o" .o); skip=0
  for f in $FILES; do [ "$b.hip" = "$f" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for f in $FILES; do
  b=$(basename "$f" .hip)
  EX=""; [ "$b" = cf_decode ] || [ "$b" = cf_loss ] || [ "$b" = cf_util ] && EX="-ffp-contract=off"
  case "$b" in *_ilp) EX="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -I"$ROOT/include" -I"$SRC" $EX $FLAGS -c "$SRC/$f" -o "$OUT/$b.o"
  OBJS="$OBJS $OUT/$b.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libcenterface_hip.so" $OBJS
echo "$OUT/libcenterface_hip.so"

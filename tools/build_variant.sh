#!/bin/bash
# Diagnostic build of the library with extra macros on ONE translation unit (timing experiments; never shipped):
#   tools/build_variant.sh exp1 -DRFD_HALO_EXP=1            ->  tools/bin/librfd_hip_exp1.so   (load with RFD_HIP_LIB=...)
#   UNIT=kernels_ring tools/build_variant.sh r1 -DRFD_RING_EXP=1   (UNIT: the csrc/*.hip file rebuilt; default kernels_conv)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
PKG="$ROOT/rs-face-detection_amd"
UNIT="${UNIT:-kernels_conv}"
mkdir -p "$ROOT/tools/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function "$@" \
    -c "${SRC:-$PKG/csrc/$UNIT.hip}" -o "$ROOT/tools/bin/${UNIT}_$NAME.o"
OBJS=""
for f in kernels_pre kernels_post kernels_conv kernels_ring kernels_f32 network detector; do
  if [ "$f" = "$UNIT" ]; then OBJS="$OBJS $ROOT/tools/bin/${UNIT}_$NAME.o"; else OBJS="$OBJS $PKG/build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/librfd_hip_$NAME.so" $OBJS -ldl
rm -f "$ROOT/tools/bin/${UNIT}_$NAME.o"
echo "built tools/bin/librfd_hip_$NAME.so"

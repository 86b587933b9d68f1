#!/bin/bash
# Diagnostic build of the library with extra macros on kernels_conv.hip (timing experiments; never shipped):
#   tools/build_variant.sh exp1 -DRFD_HALO_EXP=1   ->  tools/bin/librfd_hip_exp1.so   (load with RFD_HIP_LIB=...)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; shift
PKG="$ROOT/rs-face-detection_amd"
mkdir -p "$ROOT/tools/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-function "$@" \
    -c "${SRC:-$PKG/csrc/kernels_conv.hip}" -o "$ROOT/tools/bin/kernels_conv_$NAME.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/librfd_hip_$NAME.so" "$PKG/build/kernels_pre.o" \
    "$PKG/build/kernels_post.o" "$ROOT/tools/bin/kernels_conv_$NAME.o" "$PKG/build/kernels_f32.o" "$PKG/build/network.o" "$PKG/build/detector.o" -ldl
rm -f "$ROOT/tools/bin/kernels_conv_$NAME.o"
echo "built tools/bin/librfd_hip_$NAME.so"

#!/bin/bash
# Builds librfd_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
OBJ="$HERE/build"
mkdir -p "$OBJ"
pids=()
for f in kernels_pre kernels_post kernels_conv kernels_ring kernels_f32 network detector; do
  src="$HERE/csrc/$f.hip"
  if [ ! -f "$OBJ/$f.o" ] || [ "$src" -nt "$OBJ/$f.o" ] || [ -n "$(find "$HERE/csrc" "$HERE/../include" -name '*.h' -newer "$OBJ/$f.o")" ]; then
    $HIPCC $FLAGS -c "$src" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$HERE/librfd_hip.so" "$OBJ"/kernels_pre.o "$OBJ"/kernels_post.o "$OBJ"/kernels_conv.o "$OBJ"/kernels_ring.o "$OBJ"/kernels_f32.o "$OBJ"/network.o "$OBJ"/detector.o -ldl
echo "built $HERE/librfd_hip.so"
# compiled-language user of the C ABI through include/rfd.hpp (tests/test_cpp_facade_*.py)
g++ -std=c++17 -O2 -Wall -I"$HERE/../include" "$HERE/../tests/cpp/facade_demo.cpp" -o "$OBJ/facade_demo" -L"$HERE" -lrfd_hip -Wl,-rpath,'$ORIGIN/..' -Wl,-rpath,"$HERE" -Wl,-rpath,/opt/rocm/lib
echo "built $OBJ/facade_demo"

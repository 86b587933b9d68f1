#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the C-ABI library (aux row "race detection / sanitizers"):
# the HOST code of every .hip file is instrumented (device code is not: GPU ASan is unavailable on this pool), and the CPU
# test files that exercise the ABI without a GPU -- argument validation, error paths, the graph description calls, the weight
# converter -- run against it.  Usage: bash tools/asan_host_check.sh   (about two minutes; prints the pytest summary)
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
O=/tmp/rfd_asan; mkdir -p $O
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
pids=()
for f in kernels_pre kernels_post kernels_conv kernels_f32 network detector; do
  $HIPCC $FLAGS -c "$R/rs-face-detection_amd/csrc/$f.hip" -o $O/$f.o 2>$O/$f.err & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $O/librfd_hip_asan.so $O/kernels_pre.o $O/kernels_post.o $O/kernels_conv.o $O/kernels_f32.o $O/network.o $O/detector.o -ldl
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
cd "$R"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  RFD_HIP_LIB=$O/librfd_hip_asan.so python -m pytest tests/test_abi_cpu.py tests/test_convert_cpu.py -x -q

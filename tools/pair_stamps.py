#!/usr/bin/env python3
"""Where pw_pair_kernel's wave cycles go (diagnostic build: tools/build_variant.sh stamps -DRFD_PAIR_STAMPS; run with
RFD_HIP_LIB=tools/bin/librfd_hip_stamps.so).  Runs one op repeatedly and prints the share of each phase of the summed wave time.
usage: RFD_HIP_LIB=tools/bin/librfd_hip_stamps.so python tools/pair_stamps.py 20 [batch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import rfd_hip  # noqa: E402

op = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
det = rfd_hip.RetinaFaceDetection(max_batch_size=B, max_det=16)
det.init_synthetic_weights(1234)
g = rfd_hip.Graph()
rng = np.random.default_rng(0)
for t in range(g.num_tensors):
    td = g.tensors[t]
    if td.is_f32:
        continue
    x = np.maximum(rng.normal(0, 1, size=(1, td.height, td.width, td.channels)).astype(np.float32), 0)
    det.debug_write(t, np.repeat((x.view(np.uint32) >> 16).astype(np.uint16), B, axis=0))
L = rfd_hip.load_library()
L.rfd_debug_pair_prof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 10)()
for _ in range(5):
    det.debug_run(B, op, op)
L.rfd_debug_pair_prof(buf, 1)
N = 1   # the per-wave slots hold the LAST launch
det.set_profiling(True)
ts = []
for _ in range(N):
    det.debug_run(B, op, op)
    ts.append(float(det.op_profile(g.num_ops)[op]) * 1e3)
L.rfd_debug_pair_prof(buf, 1)
print("kernel time (HIP events, instrumented build): median %.1f us" % float(np.median(ts)))
v = np.array(list(buf), np.float64)
names = ["conv3 step: barrier wait", "conv3 step: issue_w", "conv3 step: reads + MFMA", "chunk drain vmcnt(0)", "epilogue", "conv1 step: barrier wait",
         "conv1 step: issue_w", "conv1 step: reads + MFMA", "tail drain", "wave lifetime"]
print("op %d (%s) batch %d, kernels %s: share of summed wave lifetime" % (op, g.layers[g.ops[op].layer].name.decode(), B, det.debug_op_kernels(B, op)))
for n, x in zip(names, v):
    print("  %-28s %6.1f %%   (%.0f cycles per wave per launch)" % (n, 100 * x / v[9], x / N / (8 * 200)))

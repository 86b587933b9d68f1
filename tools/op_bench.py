#!/usr/bin/env python3
"""Runs single network ops repeatedly (for rocprofv3 --pmc / --kernel-trace on one layer shape).
usage: python tools/op_bench.py --ops 27,54 --reps 20 [--batch 32] [--tile 0|1|2]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import rfd_hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ops", default="27")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--tile", type=int, default=0)
a = ap.parse_args()
det = rfd_hip.RetinaFaceDetection(max_batch_size=a.batch, max_det=16)
det.init_synthetic_weights(1234)
g = rfd_hip.Graph()
det.debug_set_conv_tile(a.tile)
rng = np.random.default_rng(0)
for t in range(g.num_tensors):  # random bf16 activations everywhere (never bench on zeros)
    td = g.tensors[t]
    if td.is_f32:
        continue
    x = np.maximum(rng.normal(0, 1, size=(1, td.height, td.width, td.channels)).astype(np.float32), 0)
    bits = (x.view(np.uint32) >> 16).astype(np.uint16)
    det.debug_write(t, np.repeat(bits, a.batch, axis=0))
for op in [int(x) for x in a.ops.split(",")]:
    o = g.ops[op]
    L = g.layers[o.layer]
    det.set_profiling(True)
    det.debug_run(a.batch, op, op)
    ts = []
    for _ in range(a.reps):
        det.debug_run(a.batch, op, op)
        ts.append(float(det.op_profile(g.num_ops)[op]) * 1e-3)
    dt = float(np.median(ts))
    fl = 2.0 * o.macs * a.batch
    print("op %d %s: %.1f us/launch (HIP events, median) %.1f TF" % (op, L.name.decode(), dt * 1e6, fl / dt / 1e12))

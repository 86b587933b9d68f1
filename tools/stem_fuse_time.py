#!/usr/bin/env python3
"""Host-timed cost of the stem + the first unit's conv1 at 16 images: as the range (0, 1) of one pass (the persistent stem kernel runs the
conv on its pooled tile) against the two ops one at a time (two launches), and the stem alone.  Every figure carries the same per-call
host overhead (debug_run synchronises), so only the differences mean anything.  usage: python tools/stem_fuse_time.py"""
import os, sys, time
sys.path.insert(0, "/root/repo/rs-face-detection_amd/python"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, rfd_hip as rfd
B = 16
det = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=B, max_det=16)
det.init_synthetic_weights(1234)
g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
tin = g.tensors[g.ops[0].in_]
x = np.random.default_rng(0).integers(0, 256, size=(B, tin.height, tin.width, tin.channels)).astype(np.float32); x[..., 3] = 0
det.debug_write(g.ops[0].in_, (x.view(np.uint32) >> 16).astype(np.uint16))
def t(f, reps=200):
    for _ in range(20): f()
    det.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    det.sync(); return (time.perf_counter() - t0) / reps * 1e6
for rnd in range(3):
    a = t(lambda: det.debug_run(B, 0, 1))
    b = t(lambda: (det.debug_run(B, 0, 0), det.debug_run(B, 1, 1)))
    c = t(lambda: det.debug_run(B, 0, 0))
    print("fused ops 0-1: %.1f us   separate: %.1f us   stem alone: %.1f us" % (a, b, c))

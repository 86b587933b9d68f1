#!/usr/bin/env python3
"""Throughput of the HOST-buffer entry point rfd_detect_batch (frames in pageable host memory, results back in
host memory): the PCIe-inclusive rate that DESIGN.md quotes next to bench.py's HBM-resident `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import rfd_hip  # noqa: E402

B = 32
det = rfd_hip.RetinaFaceDetection(max_batch_size=B, max_det=1024)
det.init_synthetic_weights(1234)
for name, hw in (("640x640", (640, 640)), ("1920x1080", (1080, 1920))):
    frames = [helpers.make_image(100 + i, *hw, n_blobs=4) for i in range(B)]
    det.call_batch(frames)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        det.call_batch(frames)
        ts.append(time.perf_counter() - t0)
    st = det.stats()
    t = float(np.median(ts))
    print("%-10s source frames, B=%d, host->host: %.2f ms/batch = %.0f img/s  (device: h2d %.2f pre %.2f net %.2f post %.2f d2h %.2f ms)" % (
        name, B, t * 1e3, B / t, st["ms_h2d"], st["ms_preprocess"], st["ms_network"], st["ms_decode"] + st["ms_sort"] + st["ms_nms"], st["ms_d2h"]))

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
    # pipelined entry (rfd_submit_batch / rfd_collect_batch): two batches in flight, frames in page-locked memory
    for label, pinned in (("pageable", False), ("page-locked", True)):
        sets = []
        for k in range(2):  # two distinct frame sets, as a producer would alternate buffers
            if pinned:
                buf = det.host_frames(B, *hw)
                for i in range(B):
                    buf[i] = frames[i]
                sets.append([buf[i] for i in range(B)])
            else:
                sets.append([f.copy() for f in frames])
        det.submit(sets[0])
        det.submit(sets[1])
        det.collect()
        det.collect()
        steps = 16
        t0 = time.perf_counter()
        det.submit(sets[0])
        for k in range(1, steps):
            det.submit(sets[k & 1])
            det.collect()
        det.collect()
        t = (time.perf_counter() - t0) / steps
        print("%-10s source frames, B=%d, pipelined submit/collect, %s frames: %.2f ms/batch = %.0f img/s" % (
            name, B, label, t * 1e3, B / t))

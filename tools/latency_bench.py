#!/usr/bin/env python3
"""p50 latency of the whole hot path at batch 1 (BASELINE.json configs[1]: RetinaFace-MobileNet-0.25 640x640 B=1;
also R50 B=1), frames resident in HBM, one synchronous rfd_detect_batch_device per sample."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402  (calibrate_cls_bias)
import helpers  # noqa: E402
import rfd_hip  # noqa: E402
from rfd_hip import parallel  # noqa: E402

dev = torch.device("cuda", 0)
for name, bb in (("RetinaFace-MobileNet0.25", rfd_hip.BACKBONE_MNET025), ("RetinaFace-R50", rfd_hip.BACKBONE_R50)):
    det = rfd_hip.RetinaFaceDetection(max_batch_size=1, max_det=1024, backbone=bb)
    det.init_synthetic_weights(1234)
    g = rfd_hip.Graph(bb, 640, 640)
    frame_np = helpers.make_image(7, 640, 640)
    bench.calibrate_cls_bias(det, g, [frame_np])   # ~100 of the 16 800 anchors clear the threshold, as in bench.py (random weights
    frame = torch.from_numpy(frame_np).to(dev)     # alone give 0 or thousands of candidates, and the NMS time of neither)
    slab = parallel.DetectionSlab(1, 1024, device=dev)
    pb, pl, pc, pt = slab.pointers()
    lat = []
    for i in range(220):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        det.detect_device([frame.data_ptr()], [(640, 640)], pb, pl, pc, pt, async_=False)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat[20:])
    st = det.stats()
    print("%-26s 640x640 B=1: p50 %.3f ms  p90 %.3f ms  (%.0f img/s)  device stages: pre %.3f net %.3f decode %.3f sort %.3f nms %.3f ms; %d kernels, %.2f GMAC; %d candidates -> %d detections" % (
        name, np.median(lat), np.quantile(lat, 0.9), 1e3 / np.median(lat), st["ms_preprocess"], st["ms_network"],
        st["ms_decode"], st["ms_sort"], st["ms_nms"], g.num_ops + 4, g.macs / 1e9, st["candidates"], int(slab.total()[0].item())))
    det.close()

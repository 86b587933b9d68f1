#!/usr/bin/env python3
"""Soak check of the timed configuration: N overlapped calls (rfd_detect_batch_device, async = 2, three slabs in rotation) on
the same 32 frames; every call's detection slab must be bit-identical to the first call's (which the oracle-checked tests
cover).  Finds rare races that a three-round test could miss.  usage: python tools/soak_check.py [calls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import helpers  # noqa: E402
import rfd_hip  # noqa: E402
from rfd_hip import parallel  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, MAX_DET = 32, 1024
det = rfd_hip.RetinaFaceDetection(max_batch_size=B, max_det=MAX_DET)
det.init_synthetic_weights(1234)
dev = torch.device("cuda", 0)
frames = [helpers.make_image(7000 + i, (640, 480, 720)[i % 3], (640, 640, 1000)[i % 3], n_blobs=5) for i in range(B)]
_, tn, _ = det.preprocess(frames[:4])
h4 = det.forward(tn)
thr = float(np.quantile(np.concatenate([h4[3 * l][:, 2:4].reshape(-1) for l in range(3)]), 0.994))
det.set_thresholds(thr, 0.45)
bufs = [torch.from_numpy(f).to(dev) for f in frames]
ptrs, shapes = [t.data_ptr() for t in bufs], [f.shape[:2] for f in frames]
slabs = [parallel.DetectionSlab(B, MAX_DET, device=dev) for _ in range(3)]
det.set_stream(torch.cuda.current_stream().cuda_stream)
det.detect_device(ptrs, shapes, *slabs[0].pointers(), async_=0)
det.sync()
ref = slabs[0].buf.clone()
assert int(slabs[0].total().sum()) > 500
bad = 0
for i in range(N):
    s = slabs[i % 3]
    det.detect_device(ptrs, shapes, *s.pointers(), async_=2)
    if i % 3 == 2:
        det.sync()
        for k, sl in enumerate(slabs):
            if not torch.equal(sl.buf, ref):
                bad += 1
                print("call %d: slab differs from the first call's (%d elements)" % (i - 2 + k, int((sl.buf != ref).sum())))
det.sync()
print("soak: %d overlapped calls, %d differing slabs, %d detections per call" % (N, bad, int(slabs[0].total().sum())))
sys.exit(1 if bad else 0)

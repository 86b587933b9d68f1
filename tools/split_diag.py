#!/usr/bin/env python3
"""diagnostic: head tensors of an unsplit pass vs the split (two chains) pass on the same frames; which images / levels differ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import helpers
import rfd_hip
from rfd_hip import parallel
B = 32
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
det = rfd_hip.RetinaFaceDetection(max_batch_size=B, max_det=1024)
det.init_synthetic_weights(1234)
det.debug_set_conv_tile(tile)
g = rfd_hip.Graph()
dev = torch.device("cuda", 0)
heads_t = [g.tensors.index(t) if False else i for i, t in enumerate(g.tensors) if t.head_level]
for k in (1,):
    frames = [helpers.make_image(7000 + 100 * k + i, (640, 480, 720)[i % 3], (640, 640, 1000)[i % 3], n_blobs=5) for i in range(B)]
    _, tn, _ = det.preprocess(frames)
    hf = det.forward(tn)                      # unsplit pass (rfd_forward)
    ref = [det.debug_read(t, B, g.tensors[t]).copy() for t in heads_t]
    bufs = [torch.from_numpy(f).to(dev) for f in frames]
    slab = parallel.DetectionSlab(B, 1024, device=dev)
    for rep in range(8):
        det.detect_device([t.data_ptr() for t in bufs], [f.shape[:2] for f in frames], *slab.pointers(), async_=0)   # split pass
        got = [det.debug_read(t, B, g.tensors[t]) for t in heads_t]
        for t, a, b in zip(heads_t, got, ref):
            d = (a != b).reshape(B, -1)
            imgs = np.nonzero(d.any(1))[0]
            if g.tensors[t].head_level == 3:
                print("set %d rep %d level 3 head: %d elements differ in images %s" % (k, rep, int(d.sum()), imgs.tolist()))
    # unsplit detect (no split) for comparison
    det.debug_set_concurrency(True, 64, 1, False)
    det.detect_device([t.data_ptr() for t in bufs], [f.shape[:2] for f in frames], *slab.pointers(), async_=0)
    got = [det.debug_read(t, B, g.tensors[t]) for t in heads_t]
    for t, a, b in zip(heads_t, got, ref):
        d = (a != b).reshape(B, -1)
        print("set %d UNSPLIT detect head tensor %d: %d elements differ in images %s" % (k, t, int(d.sum()), np.nonzero(d.any(1))[0].tolist()))
    det.debug_set_concurrency(True, 4, 2, True)

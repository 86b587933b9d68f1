"""Generates tests/golden/align_112.npz: a seeded 240x320 frame, five key points, and the 112x112 crop the CPU
oracle's FaceAlignment restatement produces (plus the 2x3 similarity).  As with heads_128.npz, the reference cannot
be run here (no rustc, no OpenCV), so the fixture pins the HIP kernel and the oracle to each other and to this
snapshot -- alignment parity is "unpinned" against real OpenCV.  Run from the repo root:
python tests/golden/make_align_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import helpers  # noqa: E402
from oracle import oracle as O  # noqa: E402

img = helpers.make_image(20241005, 240, 320, n_blobs=10)
kps, box = helpers.make_face_kps(7, 240, 320, scale_range=(1.2, 1.3))
M = O.estimate_similarity(kps, O.STANDARD_LANDMARKS)
crop, status = O.face_alignment(img, box, kps)
np.savez_compressed(os.path.join(HERE, "align_112.npz"), img=img, kps=kps, box=box, M=M, crop=crop, status=status)
print("status", status, "M", M.round(4).tolist(), "crop mean", crop.mean())

"""Generates tests/golden/heads_128.npz: seeded head tensors (net 128x128 -> 336 anchors) and the
detections the CPU oracle derives from them.  The reference cannot be run here (Rust, Triton,
OpenCV all absent), so this fixture pins the HIP path and the oracle to EACH OTHER and to this
committed snapshot, not to the reference.  Run from the repo root: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import helpers  # noqa: E402
from oracle import oracle as O  # noqa: E402

heads = [h[0] for h in helpers.make_heads(20241004, 1, 128, 128, cand_rate=0.15, n_faces=4, quantize=64)]
det_scale = np.float32(128.0 / 300.0)
det, lmk, gidx, ncand = O.decode_nms(heads, 128, 128, 0.7, 0.45, det_scale=float(det_scale))
np.savez_compressed(os.path.join(HERE, "heads_128.npz"), det=det, lmk=lmk, gidx=gidx, det_scale=det_scale,
                    ncand=ncand, **{"h%d" % i: h for i, h in enumerate(heads)})
print("candidates", ncand, "kept", len(det))

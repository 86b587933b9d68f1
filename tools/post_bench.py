#!/usr/bin/env python3
"""Times decode -> sort -> NMS in isolation on synthetic head tensors (BASELINE.json configs[4]:
dense crowds, >500 faces and thousands of candidates per image; plus the 16800-candidate worst case).
Head tensors are fed through rfd_decode_nms (host pointers); stage times are HIP-event times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import helpers  # noqa: E402
import rfd_hip  # noqa: E402

B = int(os.environ.get("B", "64"))
det = rfd_hip.RetinaFaceDetection(max_batch_size=B, max_det=2048)
for name, kw in [("typical (~100 cand/img)", dict(cand_rate=0.006, n_faces=12)),
                 ("dense crowd (600 faces)", dict(cand_rate=0.2, n_faces=600)),
                 ("very dense (cand 0.6)", dict(cand_rate=0.6, n_faces=600)),
                 ("worst case (all 16800)", dict(cand_rate=1.0))]:
    one = helpers.make_heads(3, 4, **kw)
    heads = [np.concatenate([h] * (B // 4)) for h in one]
    sc = np.full(B, 1 / 6, np.float32)
    det.decode_nms(heads, sc)
    best = None
    for _ in range(3):
        det.decode_nms(heads, sc)
        s = det.stats()
        if best is None or s["ms_decode"] + s["ms_sort"] + s["ms_nms"] < best["ms_decode"] + best["ms_sort"] + best["ms_nms"]:
            best = s
    print("%-26s B=%d cand/img=%7.0f det/img=%6.0f  decode %.3f ms  sort %.3f ms  nms %.3f ms" % (
        name, B, best["candidates"] / B, best["detections"] / B, best["ms_decode"], best["ms_sort"], best["ms_nms"]))

"""Generates tests/golden/kat_reference_inputs.json.

The reference's unit tests only print: they hold literal INPUTS but no expected
outputs.  This script takes those literal inputs (cited below) and derives the answers with plain
numpy float32 arithmetic written directly from the reference's formulas -- independently of
oracle/rfd_oracle.c, which the CPU tests then check against this file.  Run: python make_kat.py
"""
import json
import os

import numpy as np

f32 = np.float32


def iou(a, b):  # src/processing/nms.rs:39-54 (+1 convention)
    w = max(f32(0), f32(min(a[2], b[2]) - max(a[0], b[0]) + f32(1)))
    h = max(f32(0), f32(min(a[3], b[3]) - max(a[1], b[1]) + f32(1)))
    inter = f32(w * h)
    sa = f32((a[2] - a[0] + f32(1)) * (a[3] - a[1] + f32(1)))
    sb = f32((b[2] - b[0] + f32(1)) * (b[3] - b[1] + f32(1)))
    return f32(inter / f32(f32(sa + sb) - inter))


def greedy_nms(d, thr):  # nms.rs:3-65
    order = sorted(range(len(d)), key=lambda i: -d[i][4])  # python sort is stable
    keep = []
    while order:
        i = order[0]
        keep.append(i)
        order = [j for j in order[1:] if iou(d[i], d[j]) <= f32(thr)]
    return keep


def decode(boxes, deltas):  # face_detection.rs:516-549
    out = []
    for b, d in zip(boxes, deltas):
        w = f32(b[2] - b[0] + f32(1)); h = f32(b[3] - b[1] + f32(1))
        cx = f32(b[0] + f32(0.5) * f32(w - f32(1))); cy = f32(b[1] + f32(0.5) * f32(h - f32(1)))
        pcx = f32(f32(d[0] * w) + cx); pcy = f32(f32(d[1] * h) + cy)
        pw = f32(f32(np.exp(np.float64(d[2]))) * w); ph = f32(f32(np.exp(np.float64(d[3]))) * h)
        out.append([f32(pcx - f32(0.5) * f32(pw - f32(1))), f32(pcy - f32(0.5) * f32(ph - f32(1))),
                    f32(pcx + f32(0.5) * f32(pw - f32(1))), f32(pcy + f32(0.5) * f32(ph - f32(1)))])
    return out


def landmarks(boxes, deltas):  # face_detection.rs:551-570
    out = []
    for b, d in zip(boxes, deltas):
        w = f32(b[2] - b[0] + f32(1)); h = f32(b[3] - b[1] + f32(1))
        cx = f32(b[0] + f32(0.5) * f32(w - f32(1))); cy = f32(b[1] + f32(0.5) * f32(h - f32(1)))
        out.append([f32(f32(d[k] * (w if k % 2 == 0 else h)) + (cx if k % 2 == 0 else cy)) for k in range(10)])
    return out


def base_anchors(base_size, scales):  # generate_anchors.rs:61-93 with ratio 1
    s = f32(base_size)
    ctr = f32(0.5) * f32(s - f32(1))
    ws = f32(np.round(np.sqrt(f32(s * s))))
    out = []
    for sc in scales:
        w = f32(ws * f32(sc))
        out.append([f32(ctr - f32(0.5) * f32(w - 1)), f32(ctr - f32(0.5) * f32(w - 1)),
                    f32(ctr + f32(0.5) * f32(w - 1)), f32(ctr + f32(0.5) * f32(w - 1))])
    return out


def geometry(h, w, sw=640, sh=640):  # face_detection.rs:140-153
    r = f32(h) / f32(w); m = f32(sh) / f32(sw)
    if r > m:
        nh = sh; nw = int(f32(nh) / r)
    else:
        nw = sw; nh = int(f32(nw) * r)
    return nw, nh, float(f32(nh) / f32(h))


def tolist(x):
    return [[float(v) for v in r] for r in x]


nms_boxes = [[100.0, 100.0, 210.0, 210.0, 0.72], [250.0, 250.0, 420.0, 420.0, 0.8],
             [220.0, 220.0, 320.0, 330.0, 0.92], [100.0, 100.0, 210.0, 210.0, 0.6]]
nb = [[f32(v) for v in r] for r in nms_boxes]
dec_boxes = [[50.0, 50.0, 150.0, 150.0], [30.0, 30.0, 200.0, 200.0]]
dec_deltas = [[0.1, 0.2, 0.1, 0.2], [0.2, 0.1, 0.2, 0.1]]
lmk_deltas = [[0.1, 0.2, 0.1, 0.2, 0.2, 0.1, 0.2, 0.1, 0.3, 0.3],
              [0.2, 0.1, 0.2, 0.1, 0.1, 0.2, 0.1, 0.2, 0.3, 0.3]]
clip_in = [[50.0, 50.0, 150.0, 150.0, 60.0, 60.0, 160.0, 160.0], [30.0, 30.0, 200.0, 200.0, 40.0, 40.0, 220.0, 220.0]]
kat = {
    "_comment": "inputs: literal arrays of the reference's print-only unit tests; answers: derived by "
                "tests/golden/make_kat.py from the reference's formulas (hand-checked in SURVEY.md Appendix C)",
    "nms": {"source": "src/processing/nms.rs:76-83 (thr 0.4) and src/rcnn/cpu_nms.rs:65-72 (thr 0.3)",
            "boxes": nms_boxes, "thr": [0.4, 0.3],
            "keep": [greedy_nms(nb, 0.4), greedy_nms(nb, 0.3)],
            "iou_2_1": float(iou(nb[2], nb[1])), "iou_2_0": float(iou(nb[2], nb[0])), "iou_0_3": float(iou(nb[0], nb[3]))},
    "anchor_plane": {"source": "src/rcnn/anchors.rs:30-39", "base": [[0.0, 0.0, 15.0, 15.0], [0.0, 0.0, 31.0, 31.0]],
                     "height": 2, "width": 2, "stride": 16,
                     "out": [[[[b[0] + 16 * w, b[1] + 16 * h, b[2] + 16 * w, b[3] + 16 * h]
                               for b in ([0.0, 0.0, 15.0, 15.0], [0.0, 0.0, 31.0, 31.0])] for w in range(2)] for h in range(2)]},
    "anchors_fpn": {"source": "src/processing/generate_anchors.rs:217-250 == face_detection.rs:55-80",
                    "out": [tolist(base_anchors(16, (32, 16))), tolist(base_anchors(16, (8, 4))), tolist(base_anchors(16, (2, 1)))]},
    "bbox_pred": {"source": "src/processing/bbox_transform.rs:253-260 (first 4 delta columns; formula face_detection.rs:516-549)",
                  "boxes": dec_boxes, "deltas": dec_deltas,
                  "out": tolist(decode([[f32(v) for v in r] for r in dec_boxes], [[f32(v) for v in r] for r in dec_deltas]))},
    "landmark_pred": {"source": "src/processing/bbox_transform.rs:268-275 (formula face_detection.rs:551-570)",
                      "boxes": dec_boxes, "deltas": lmk_deltas,
                      "out": tolist(landmarks([[f32(v) for v in r] for r in dec_boxes], [[f32(v) for v in r] for r in lmk_deltas]))},
    "clip_boxes": {"source": "src/processing/bbox_transform.rs:213-217 (im_shape 100x100)", "boxes": clip_in, "im_shape": [100, 100],
                   "out": [[min(max(v, 0.0), 99.0) for v in r] for r in clip_in]},
    "geometry": {"source": "face_detection.rs:140-153, image_size (640,640)",
                 "cases": [{"h": h, "w": w, "out": list(geometry(h, w))}
                           for (h, w) in [(1080, 1920), (2160, 3840), (768, 1024), (640, 480), (480, 641), (1280, 1280), (100, 100), (479, 641)]]},
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_reference_inputs.json"), "w") as f:
    json.dump(kat, f, indent=1)
print("wrote kat_reference_inputs.json")

#!/usr/bin/env python3
"""Diagnostic: where does conv3x3_halo_kernel (force_tile 13 / 14) differ from the merged-kx kernel (7) on one op?
usage: python tools/halo_diag.py OP [N] [TILE]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import rfd_hip as rfd  # noqa: E402

op = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 16; tile = int(sys.argv[3]) if len(sys.argv) > 3 else 13
det = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=n, max_det=16)
det.init_synthetic_weights(4321)
g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
o = g.ops[op]; L = g.layers[o.layer]
rng = np.random.default_rng(100 + n)
td = g.tensors[o.in_]; to = g.tensors[o.out]
x = np.maximum(rng.normal(0, 1, size=(n, td.height, td.width, td.channels)), 0).astype(np.float32)
det.debug_write(o.in_, (x.view(np.uint32) >> 16).astype(np.uint16))
f = lambda a: (a.astype(np.uint32) << 16).view(np.float32)
res = {}
for t in (7, tile):
    det.debug_set_conv_tile(t)
    det.debug_write(o.out, np.full((n, to.height, to.width, to.channels), 0x7fc0, np.uint16))
    det.debug_run(n, op, op)
    res[t] = f(det.debug_read(o.out, n, to))[..., o.y_coff:o.y_coff + L.cout]
a, b = res[tile], res[7]
d = np.abs(a - b); tol = np.maximum(np.abs(a), np.abs(b)) * 2.0 ** -7
bad = np.argwhere(~(d <= tol))
print(L.name.decode(), "shape", a.shape, "identical frac", float((a == b).mean()), "bad", len(bad), "nan", int(np.isnan(a).sum()))
print("value range", float(np.nanmin(b)), float(np.nanmax(b)), "abs mean", float(np.abs(b).mean()))
for r in bad[:40]:
    print(tuple(int(v) for v in r), "halo", a[tuple(r)], "kx", b[tuple(r)])

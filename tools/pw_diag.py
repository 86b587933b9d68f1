#!/usr/bin/env python3
"""diagnostic: persistent pointwise kernel vs generic kernel on one op at batch n: where do they differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import rfd_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ops = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "25").split(",")]
det = rfd_hip.RetinaFaceDetection(max_batch_size=32, max_det=16)
det.init_synthetic_weights(1234)
g = rfd_hip.Graph()
rng = np.random.default_rng(1)
for i in ops:
    o = g.ops[i]
    for t in (o.in_, o.res):
        td = g.tensors[t]
        x = rng.normal(0, 1, size=(n, td.height, td.width, td.channels)).astype(np.float32)
        det.debug_write(t, (x.view(np.uint32) >> 16).astype(np.uint16))
    outs = [t for t in (o.out, o.out2) if t >= 0]
    def run(tile):
        det.debug_set_conv_tile(tile)
        for t in outs:
            td = g.tensors[t]
            det.debug_write(t, np.full((n, td.height, td.width, td.channels), 0x7fc0, np.uint16))
        det.debug_run(n, i, i)
        return [det.debug_read(t, n, g.tensors[t]).reshape(-1, g.tensors[t].channels) for t in outs]
    ref = run(7)
    for tile in (0, 0, 0, 0, 0, 0):
        got = run(tile)
        for k, (a, b) in enumerate(zip(got, ref)):
            d = a != b
            rows = np.nonzero(d.any(1))[0]
            cols = np.nonzero(d.any(0))[0]
            msg = "op %d tile %d out%d: %d / %d elements differ" % (i, tile, k, int(d.sum()), d.size)
            if len(rows):
                tiles = np.unique(rows // 128)
                msg += "; rows %d..%d (%d rows, tiles %s%s), row%%128 hist16 %s, channel chunks %s, col%%64 hist8 %s" % (
                    rows[0], rows[-1], len(rows), tiles[:12].tolist(), "..." if len(tiles) > 12 else "",
                    np.bincount((rows % 128) // 16, minlength=8).tolist(), np.unique(cols // 128).tolist(),
                    np.bincount((cols % 64) // 8, minlength=8).tolist())
                nan = int(((a[d] & 0x7fff) > 0x7f80).sum())
                msg += "; NaN among them %d" % nan
            print(msg)

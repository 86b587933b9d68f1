#!/usr/bin/env python3
"""Per-layer timing of every implicit-GEMM op under each forced tile configuration (rfd_debug_set_conv_tile):
which layers would gain from a different tile than the launch heuristic picks.  usage: tile_sweep.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import rfd_hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tiles = [int(t) for t in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,3,4".split(","))]
det = rfd_hip.RetinaFaceDetection(max_batch_size=B, max_det=16)
det.init_synthetic_weights(1234)
det.debug_set_concurrency(False, 8, 1, False)
g = rfd_hip.Graph()
rng = np.random.default_rng(0)
for t in range(g.num_tensors):
    td = g.tensors[t]
    if td.is_f32:
        continue
    x = np.maximum(rng.normal(0, 1, size=(1, td.height, td.width, td.channels)).astype(np.float32), 0)
    det.debug_write(t, np.repeat((x.view(np.uint32) >> 16).astype(np.uint16), B, axis=0))
ops = [k for k, o in enumerate(g.ops) if o.kind == 2]
res = {}
det.set_profiling(True)
for _ in range(30):  # clocks up before anything is timed
    det.debug_run(B, 0, -1)
for rnd in range(2):
  for tile in tiles:
    det.debug_set_conv_tile(tile)
    for op in ops:
        det.debug_run(B, op, op)
        ts = []
        for _ in range(7):
            det.debug_run(B, op, op)
            ts.append(float(det.op_profile(g.num_ops)[op]) * 1e3)
        res[(tile, op)] = min(res.get((tile, op), 1e9), float(np.median(ts)))
print("batch", B, "us per launch by forced tile", tiles)
tot = {t: 0.0 for t in tiles}
best_tot = 0.0
for op in ops:
    o = g.ops[op]
    L = g.layers[o.layer]
    row = [res[(t, op)] for t in tiles]
    b = int(np.argmin(row))
    best_tot += row[b]
    for t, v in zip(tiles, row):
        tot[t] += v
    print("%3d %-22s k%d %4d->%4d  " % (op, L.name.decode(), L.kh, L.cin, L.cout) + "  ".join("%7.1f" % v for v in row) +
          ("   best=tile%d (%.0f%%)" % (tiles[b], 100 * (row[0] - row[b]) / row[0]) if b != 0 and row[b] < 0.97 * row[0] else ""))
print("totals", {t: round(v, 1) for t, v in tot.items()}, "per-op best", round(best_tot, 1))

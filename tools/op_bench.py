#!/usr/bin/env python3
"""Times single network ops (HIP events around the launch, median of --reps runs) under forced conv tile configurations
(rfd_debug_set_conv_tile) and batch sizes, interleaved in ONE process (A/B rule: never compare across processes).
usage: python tools/op_bench.py --ops 13,25 --tiles 0,7 --batches 32,16 [--reps 20] [--rounds 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import rfd_hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ops", default="27")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--batches", default="32")
ap.add_argument("--tiles", default="0")
a = ap.parse_args()
batches = [int(x) for x in a.batches.split(",")]
tiles = [int(x) for x in a.tiles.split(",")]
det = rfd_hip.RetinaFaceDetection(max_batch_size=max(batches), max_det=16)
det.init_synthetic_weights(1234)
g = rfd_hip.Graph()
rng = np.random.default_rng(0)
for t in range(g.num_tensors):  # random bf16 activations everywhere (never bench on zeros)
    td = g.tensors[t]
    if td.is_f32:
        continue
    x = np.maximum(rng.normal(0, 1, size=(1, td.height, td.width, td.channels)).astype(np.float32), 0)
    bits = (x.view(np.uint32) >> 16).astype(np.uint16)
    det.debug_write(t, np.repeat(bits, max(batches), axis=0))
det.set_profiling(True)
for _ in range(20):  # clocks up
    det.debug_run(max(batches), 0, -1)
if a.ops == "all":
    ops = [k for k, o in enumerate(g.ops) if o.kind in (2, 6)]
elif a.ops == "halo":  # the layers conv3x3_halo_kernel accepts
    ops = [k for k, o in enumerate(g.ops) if o.kind == 2 and g.layers[o.layer].kh == 3 and g.layers[o.layer].stride == 1
           and g.layers[o.layer].cin % 128 == 0 and g.layers[o.layer].cout % 128 == 0 and g.layers[o.layer].cout <= 512
           and o.res < 0 and o.layer_n2 < 0 and g.tensors[o.in_].width in (80, 40)]
elif a.ops == "pwg":  # the layers pw_gemm_kernel accepts
    ops = [k for k, o in enumerate(g.ops) if o.kind == 2 and g.layers[o.layer].kh == 1 and g.layers[o.layer].stride == 1
           and o.layer2 < 0 and o.out2 < 0 and o.outf < 0 and g.layers[o.layer].cin % 128 == 0
           and g.layers[o.layer].cin >= 256 and g.layers[o.layer].cout % 128 == 0 and g.layers[o.layer].cout <= 1024]
elif a.ops == "pww":  # the layers pw_wide_kernel accepts
    ops = [k for k, o in enumerate(g.ops) if o.kind == 2 and g.layers[o.layer].kh == 1 and g.layers[o.layer].stride == 1
           and o.in_affine < 0 and o.outf < 0 and not o.res_up2 and not o.res_post and g.layers[o.layer].cout % 256 == 0
           and 512 <= g.layers[o.layer].cout <= 2048
           and g.layers[o.layer].cin + (g.layers[o.layer2].cin if o.layer2 >= 0 else 0) >= 384]
else:
    ops = [int(x) for x in a.ops.split(",")]
res = {}
for rnd in range(a.rounds):
    for B in batches:
        for tile in tiles:
            det.debug_set_conv_tile(tile)
            for op in ops:
                if det.debug_op_kernels(B, op)[0].startswith("(fused"):   # runs inside the previous op's kernel in a real pass
                    res.setdefault((B, tile, op), []).append(0.0)
                    continue
                det.debug_run(B, op, op)
                ts = []
                for _ in range(a.reps):
                    det.debug_run(B, op, op)
                    ts.append(float(det.op_profile(g.num_ops)[op]) * 1e3)
                res.setdefault((B, tile, op), []).append(float(np.median(ts)))
for B in batches:
    print("batch %d: us per launch (median over %d rounds of medians of %d), tiles %s" % (B, a.rounds, a.reps, tiles))
    tot = {t: 0.0 for t in tiles}
    for op in ops:
        o = g.ops[op]
        L = g.layers[o.layer]
        row = [float(np.median(res[(B, t, op)])) for t in tiles]
        for t, v in zip(tiles, row):
            tot[t] += v
        fl = 2.0 * o.macs * B
        print("%3d %-22s k%d %4d->%4d  " % (op, L.name.decode(), L.kh, L.cin, L.cout) +
              "  ".join(("%7.1f us %6.0f TF" % (v, fl / v / 1e6)) if v > 0 else "    (fused into the stem)" for v in row))
    print("    totals", {t: round(v, 1) for t, v in tot.items()})

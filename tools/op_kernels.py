#!/usr/bin/env python3
"""Asks the LIBRARY which kernel(s) run every op of the network (rfd_debug_op_kernels: launch_conv()'s own decision on this
device, nothing launched) and writes {op index: {"layer": name, "kernels": [...]}} as JSON -- the op -> kernel map that
tools/traffic_model.py and tools/roof_gap.py attribute bytes and time with (round-3 review: their Python mirror of launch_conv()
had gone stale).  Needs the GPU (a context does).   usage: python tools/op_kernels.py [--batch 16] [--solo] > op_kernels.json"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rs-face-detection_amd", "python"))
import rfd_hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16, help="images per chain (the timed mode runs two chains of 16)")
ap.add_argument("--solo", action="store_true", help="a chain that has the GPU to itself (unsplit pass) instead of one of two")
a = ap.parse_args()
det = rfd_hip.RetinaFaceDetection(max_batch_size=a.batch, max_det=16)
det.init_synthetic_weights(1234)
g = rfd_hip.Graph()
out = {"batch_per_chain": a.batch, "co_running": not a.solo, "ops": {}}
for i, o in enumerate(g.ops):
    out["ops"][str(i)] = {"layer": g.layers[o.layer].name.decode(), "kind": o.kind, "kernels": det.debug_op_kernels(a.batch, i, not a.solo)}
json.dump(out, sys.stdout, indent=1)
print()

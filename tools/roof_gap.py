#!/usr/bin/env python3
"""Per op (B = 16 section of a tools/op_bench.py table): time next to its two floors -- MFMA at 2.5 PF/s and HBM at 5 TB/s
(algorithmic bytes: every tensor the op reads or writes once) -- sorted by the time above the larger floor.  Host only.
The kernel column is the library's own op -> kernel answer (tools/op_kernels.py -> JSON, optional second argument).
usage: python tools/roof_gap.py profiles/rNN_per_op_b16_b32.txt [profiles/rNN_op_kernels_b16.json]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rs-face-detection_amd", "python"))
import rfd_hip  # noqa: E402

B = 16
g = rfd_hip.Graph(rfd_hip.BACKBONE_R50, 640, 640)
kmap = json.load(open(sys.argv[2]))["ops"] if len(sys.argv) > 2 else {}
times = {}
sec = None
for ln in open(sys.argv[1]):
    m = re.match(r"batch (\d+):", ln)
    if m:
        sec = int(m.group(1))
        continue
    m = re.match(r"\s*(\d+)\s+(\S+)\s+k\d\s+\d+->\s*\d+\s+([\d.]+) us", ln)
    if m and sec == B:
        times[int(m.group(1))] = float(m.group(3))


def tb(t, ch=None):
    if t < 0:
        return 0
    T = g.tensors[t]
    return (ch if ch else T.channels) * T.height * T.width * (4 if T.is_f32 else 2) * B


rows = []
for i, us in times.items():
    o = g.ops[i]
    L = g.layers[o.layer]
    n_out = L.cout + (g.layers[o.layer_n2].cout if o.layer_n2 >= 0 else 0)
    rd = tb(o.in_, max(L.cin, 64) if g.tensors[o.in_].channels > max(L.cin, 64) else None) + (tb(o.in2) // (g.layers[o.layer2].stride ** 2) if o.layer2 >= 0 else 0) + tb(o.res)  # a stride-2 shortcut reads a quarter of its input
    wr = sum(tb(t, n_out if (t == o.out and g.tensors[t].channels > n_out) else None) for t in (o.out, o.out2, o.outf, o.out_b) if t >= 0)
    t_mfma = 2.0 * o.macs * B / 2.5e15 * 1e6
    t_hbm = (rd + wr) / 5.0e12 * 1e6
    floor = max(t_mfma, t_hbm)
    rows.append((us - floor, i, L.name.decode(), us, t_mfma, t_hbm, "mfma" if t_mfma > t_hbm else "hbm"))
rows.sort(reverse=True)
tot = sum(r[3] for r in rows)
print("%3s %-22s %8s %8s %8s %5s %8s  %s   (B = %d, sum %.0f us)" % ("op", "layer", "us", "mfma us", "hbm us", "bound", "above", "kernel", B, tot))
for above, i, nm, us, tm, th, bd in rows:
    print("%3d %-22s %8.1f %8.1f %8.1f %5s %8.1f  %s" % (i, nm, us, tm, th, bd, above, " + ".join(kmap.get(str(i), {}).get("kernels", []))))
print("sum of floors %.0f us" % sum(max(r[4], r[5]) for r in rows))

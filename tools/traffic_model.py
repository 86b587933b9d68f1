#!/usr/bin/env python3
"""Algorithmic HBM bytes per forward pass (B = 32) per kernel class -- every tensor an op reads / writes counted once
(weights excluded: 54 MB per pass in total) -- next to the PMC traffic of profiles/*hbm_traffic.json.  The op -> kernel
mapping is the library's own (tools/op_kernels.py -> JSON, second argument).  Host only.
usage: python tools/traffic_model.py profiles/rNN_hbm_traffic.json profiles/rNN_op_kernels_b16.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rs-face-detection_amd", "python"))
import rfd_hip  # noqa: E402

B = 32
g = rfd_hip.Graph(rfd_hip.BACKBONE_R50, 640, 640)
pm = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")))


def tbytes(t, ch=None):
    if t < 0:
        return 0
    T = g.tensors[t]
    return (ch if ch else T.channels) * T.height * T.width * (4 if T.is_f32 else 2) * B


# op -> kernel: the library's own answer (tools/op_kernels.py on the GPU box -> profiles/rNN_op_kernels_b16.json), not a mirror of
# launch_conv()'s rules
kmap = json.load(open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "op_kernels_latest.json")))["ops"]


def fused_into(i):
    """the op runs inside the previous op's kernel in a real pass (the stem with the first conv1): its input is not read back"""
    return kmap[str(i)]["kernels"][0].startswith("(fused into ")


def op_class(i):
    ks = kmap[str(i)]["kernels"]
    if fused_into(i):
        return "rfd::" + ks[0][len("(fused into "):].rstrip(")").split("<")[0]
    return "rfd::" + ks[0].split("<")[0] if len(ks) == 1 else " + ".join("rfd::" + k.split("<")[0] for k in ks)


cls = {}
for i, o in enumerate(g.ops):
    L = g.layers[o.layer]
    n_out = L.cout + (g.layers[o.layer_n2].cout if o.layer_n2 >= 0 else 0)
    if o.kind not in (2, 3, 6):
        continue
    k = op_class(i).replace("conv_b2b_s1_persistent_k128_kernel", "conv_b2b_s1_kernel").replace("conv_b2b_s1_persistent_kernel", "conv_b2b_s1_kernel")
    rd = tbytes(o.in_, max(L.cin, 64) if (o.kind == 2 and g.tensors[o.in_].channels > max(L.cin, 64)) else None) + (tbytes(o.in2) // (g.layers[o.layer2].stride ** 2) if o.layer2 >= 0 else 0) + tbytes(o.res)  # a stride-2 shortcut reads every other pixel of every other row
    if fused_into(i):
        rd = 0
    wr = 0
    for t in (o.out, o.out2, o.outf, o.out_b):
        if t >= 0:
            T = g.tensors[t]
            wr += tbytes(t, n_out if (t == o.out and T.channels > n_out and T.channels_logical != L.cout) else None)
    a = cls.setdefault(k, [0, 0, 0])
    a[0] += rd
    a[1] += wr
    a[2] += 1
meas = {}
for name, v in pm["kernels"].items():
    base = name.split("<")[0].replace("conv_b2b_s1_persistent_k128_kernel", "conv_b2b_s1_kernel").replace("conv_b2b_s1_persistent_kernel", "conv_b2b_s1_kernel")
    m = meas.setdefault(base, [0.0, 0.0])
    m[0] += v["fetch_bytes_per_pass"]
    m[1] += v["write_bytes_per_pass"]
print("%-28s %4s %10s %10s %10s %10s %7s" % ("kernel class", "ops", "alg rd MB", "PMC rd MB", "alg wr MB", "PMC wr MB", "PMC/alg"))
tot = [0, 0, 0, 0]
for k in sorted(cls, key=lambda k: -(cls[k][0] + cls[k][1])):
    rd, wr, n = cls[k]
    f, w = meas.get(k, [0, 0])
    tot = [tot[0] + rd, tot[1] + f, tot[2] + wr, tot[3] + w]
    print("%-28s %4d %10.0f %10.0f %10.0f %10.0f %7.2f" % (k.replace("rfd::", ""), n, rd / 1e6, f / 1e6, wr / 1e6, w / 1e6, (f + w) / (rd + wr)))
print("%-28s %4s %10.0f %10.0f %10.0f %10.0f %7.2f" % ("total", "", tot[0] / 1e6, tot[1] / 1e6, tot[2] / 1e6, tot[3] / 1e6, (tot[1] + tot[3]) / (tot[0] + tot[2])))

#!/usr/bin/env python3
"""Algorithmic HBM bytes per forward pass (B = 32) per kernel class -- every tensor an op reads / writes counted once
(weights excluded: 54 MB per pass in total) -- next to the PMC traffic of profiles/*hbm_traffic.json.  The op -> kernel
mapping mirrors launch_conv() for the split (co-running) mode.  Host only."""
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


BC = 16  # images per chain in the timed (split) mode: the size rules of launch_conv() see this batch


def conv_class(o, L, n_out):
    """The kernel launch_conv() picks for this op at BC images (mirrors its order of tests)."""
    tin = g.tensors[o.in_]
    tout = g.tensors[o.out if o.out >= 0 else (o.out2 if o.out2 >= 0 else o.outf)]
    M = BC * tout.height * tout.width
    plain_out = o.outf < 0 and o.layer_b < 0
    if L.kh == 1 and L.stride == 1 and o.res >= 0 and o.layer2 < 0 and o.in_affine < 0 and L.cin in (64, 128, 256) and L.cout >= 4 * L.cin \
            and L.cout <= 1024 and not o.res_up2 and not o.res_post and plain_out and M >= 128 * 128:
        return "rfd::pw_stream_kernel"
    if L.kh == 1 and L.stride == 1 and o.layer2 < 0 and o.out2 < 0 and plain_out and L.cin % 128 == 0 and 256 <= L.cin <= 2048 \
            and L.cout % 128 == 0 and L.cout <= 1024 and -(-M // 256) * (L.cout // 128) >= (400 if o.res >= 0 else 150):
        return "rfd::pw_gemm_kernel"
    kk = L.cin + (g.layers[o.layer2].cin if o.layer2 >= 0 else 0)
    if L.kh == 1 and L.stride == 1 and o.in_affine < 0 and plain_out and kk >= 384 and L.cout % 256 == 0 and 512 <= L.cout <= 2048 \
            and not o.res_up2 and not o.res_post and -(-M // 256) * (L.cout // 256) >= 150:
        return "rfd::pw_wide_kernel"
    if L.kh == 3 and L.stride == 1 and L.cin == 64 and L.cout == 64 and o.layer_n2 < 0 and M >= 96 * 256:
        return "rfd::conv3x3_c64_kernel"
    if L.kh == 3 and L.stride == 1 and L.cin % 128 == 0 and (n_out % 128 == 0 or n_out == 192) and n_out <= 512 and o.res < 0 \
            and (tin.width % 16 == 0 or tin.width == 40):
        tiles = BC * (-(-tin.height // 6) if tin.width == 40 else (tin.width // 16) * -(-tin.height // 16))
        if (n_out == 192 and tiles >= 100) or (n_out != 192 and tiles * (n_out // 128) >= 200):
            return "rfd::conv3x3_halo_kernel"
    if L.kh == 3 and L.stride == 1 and n_out % 128 == 0 and o.layer2 < 0:
        return "rfd::conv3x3_kx_kernel"
    return "rfd::conv_igemm_kernel"


cls = {}
for i, o in enumerate(g.ops):
    L = g.layers[o.layer]
    n_out = L.cout + (g.layers[o.layer_n2].cout if o.layer_n2 >= 0 else 0)
    if o.kind == 3:
        k = "rfd::stem_kernel"
    elif o.kind == 6:
        Lb = g.layers[o.layer_b]
        # stage 1's own kernels / round 3's pair kernels (stage 3: the register-operand form)
        k = "rfd::conv_b2b_s1_kernel" if (L.cin == 64 and Lb.cout == 64) else ("rfd::pw_pair_kernel" if L.cin == 256 else "rfd::pw_b2b_kernel")
    elif o.kind != 2:
        continue
    else:
        k = conv_class(o, L, n_out)
    rd = tbytes(o.in_, max(L.cin, 64) if (o.kind == 2 and g.tensors[o.in_].channels > max(L.cin, 64)) else None) + tbytes(o.in2) + tbytes(o.res)
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

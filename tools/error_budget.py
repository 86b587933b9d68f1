#!/usr/bin/env python3
"""Where the bf16 product path loses its accuracy: the network walked on the CPU (tests/torch_ref.py: the library's own op list,
torch f32 convolutions) with bf16 rounding switched on for ONE group of ops at a time -- stem, stages 1-4, FPN, the SSH modules,
the heads -- or for the weights only, against the walk with no rounding at all.  Per variant: the fg-logit error of the anchors
near the 0.7 threshold (0.5 < p < 0.95 in the unrounded walk; mean / p99), the candidate flips at the threshold and the relative
L2 error of the box deltas.  The f32 parameters come from a context in the f32 parity mode (unrounded), the op list and the
rounding points from the graph; random calibrated weights as in tests/test_t2_gpu.py (no checkpoint exists: the figures are an
upper bound for a trained detector, whose scores do not crowd the threshold).  Needs the GPU box only to read the parameters back.
usage: python tools/error_budget.py [--frames 4] > gpurun_out/error_budget.txt"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import helpers  # noqa: E402
import rfd_hip  # noqa: E402
import torch_ref  # noqa: E402
import unfolded_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402
from rfd_hip import convert  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4)
a = ap.parse_args()
THR = 0.7
N = a.frames
g = rfd_hip.Graph(rfd_hip.BACKBONE_R50, 640, 640)
P = unfolded_ref.make_params(777)
frames = [helpers.make_image(9000 + i, 640, 640, n_blobs=8) for i in range(N)]
tensor = np.stack([O.preprocess(f, 640, 640)[1] for f in frames])
# the calibration of tests/test_t2_gpu.py (head gains + cls bias so that ~0.6 % of the anchors clear 0.7)
with torch.no_grad():
    h0 = unfolded_ref.forward(P, torch.from_numpy(tensor[:4]))
for l, st in enumerate((32, 16, 8)):
    pr = np.clip(h0[3 * l][:, 2:4].astype(np.float64), 1e-12, 1 - 1e-12)
    gains = {"cls": 1.5 / float(np.std(np.log(pr / (1 - pr)))), "bbox": 0.3 / float(np.std(h0[3 * l + 1])), "lmk": 0.4 / float(np.std(h0[3 * l + 2]))}
    for k, gk in gains.items():
        P["head%d_%s_weight" % (st, k)] = (P["head%d_%s_weight" % (st, k)] * np.float32(gk)).astype(np.float32)
        P["head%d_%s_bias" % (st, k)] = (P["head%d_%s_bias" % (st, k)] * np.float32(gk)).astype(np.float32)
with torch.no_grad():
    h0 = unfolded_ref.forward(P, torch.from_numpy(tensor[:4]))
p = np.clip(np.concatenate([h0[3 * l][:, 2:4].reshape(-1) for l in range(3)]).astype(np.float64), 1e-12, 1 - 1e-12)
delta = float(np.log(THR / (1 - THR)) - np.quantile(np.log(p / (1 - p)), 1.0 - 0.006))
for st in (32, 16, 8):
    P["head%d_cls_bias" % st] = P["head%d_cls_bias" % st].copy()
    P["head%d_cls_bias" % st][2:4] += np.float32(delta)

det = rfd_hip.RetinaFaceDetection(max_batch_size=1, max_det=16, precision=rfd_hip.PRECISION_F32)
convert.import_unfolded(det, g, P)
ref = torch_ref.TorchRef(g, det, round_bf16=False)       # unrounded folded f32 parameters
w32 = [w.clone() for w in ref.w]
wbf = [torch.from_numpy(helpers.bf16_round(w.numpy())) for w in w32]
det.close()


def group_of(i):
    n = g.layers[g.ops[i].layer].name.decode()
    for pre in ("stage1", "stage2", "stage3", "stage4", "fpn", "ssh", "head"):
        if n.startswith(pre):
            return pre
    return "stem"


groups = ["stem", "stage1", "stage2", "stage3", "stage4", "fpn", "ssh", "head"]
ops_of = {k: {i for i in range(len(g.ops)) if group_of(i) == k} for k in groups}
x4 = torch.cat([torch.from_numpy(tensor), torch.zeros(N, 1, 640, 640)], 1)


def walk(weights, round_ops):
    ref.w = weights
    ref.round_ops = round_ops
    return ref.heads(ref.forward(x4))


def fg(heads):
    return np.concatenate([np.transpose(heads[3 * l][:, 2:4], (0, 2, 3, 1)).reshape(N, -1) for l in range(3)], 1).astype(np.float64)


def logit(pv):
    pv = np.clip(pv, 1e-9, 1 - 1e-9)
    return np.log(pv / (1 - pv))


base = walk(w32, set())
fb = fg(base)
near = (fb > 0.5) & (fb < 0.95)
cand = fb >= THR
rows = []
variants = [("weights only (bf16 weights, f32 activations)", wbf, set())] + \
           [("activations of %s only" % k, w32, ops_of[k]) for k in groups] + \
           [("all activations, f32 weights", w32, set(range(len(g.ops)))),
            ("everything (the product path's arithmetic)", wbf, set(range(len(g.ops)))),
            ("everything except ssh + head activations", wbf, set(range(len(g.ops))) - ops_of["ssh"] - ops_of["head"]),
            ("everything except stage 3-4 + fpn + ssh + head activations", wbf, ops_of["stem"] | ops_of["stage1"] | ops_of["stage2"])]
print("bf16 error budget: %d frames, %d anchors near the threshold, %d candidates >= %.1f in the unrounded walk" % (N, int(near.sum()), int(cand.sum()), THR))
print("%-62s %10s %10s %10s %8s %10s" % ("rounding applied to", "logit mean", "logit p99", "logit max", "flips", "bbox relL2"))
out = []
for name, w, ro in variants:
    h = walk(w, ro)
    f = fg(h)
    le = np.abs(logit(f[near]) - logit(fb[near]))
    flips = int(((f >= THR) != cand).sum())
    bb = np.concatenate([h[3 * l + 1].ravel() for l in range(3)]); b0 = np.concatenate([base[3 * l + 1].ravel() for l in range(3)])
    rel = float(np.linalg.norm(bb - b0) / np.linalg.norm(b0))
    print("%-62s %10.4f %10.4f %10.4f %8d %10.2e" % (name, le.mean(), np.percentile(le, 99), le.max(), flips, rel))
    out.append({"variant": name, "fg_logit_abs_err_mean": float(le.mean()), "p99": float(np.percentile(le, 99)), "max": float(le.max()),
                "threshold_flips": flips, "flip_rate": flips / max(int(cand.sum()), 1), "bbox_rel_l2": rel})
print("(independent error sources add in quadrature: sqrt(sum of the single-group means^2) = %.4f)" %
      float(np.sqrt(sum(o["fg_logit_abs_err_mean"] ** 2 for o in out[:9]))))
try:
    json.dump({"frames": N, "near_threshold_anchors": int(near.sum()), "candidates": int(cand.sum()), "variants": out},
              open(os.path.join(ROOT, "gpurun_out", "error_budget.json"), "w"), indent=1)
except OSError:
    pass

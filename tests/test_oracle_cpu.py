"""CPU tests: the oracle (oracle/rfd_oracle.c) against the known answers derived from the literal
inputs of the reference's own print-only tests (tests/golden/kat_reference_inputs.json) and against
size-independent properties of the reference algorithm.  Parity status of the oracle itself:
"parity unpinned" (the reference holds no expected outputs) -- see DESIGN.md."""
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import helpers

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_reference_inputs.json")))


def test_nms_kat(oracle):
    k = KAT["nms"]
    for thr, keep in zip(k["thr"], k["keep"]):
        assert oracle.nms(np.array(k["boxes"], np.float32), thr).tolist() == keep


def test_anchor_kats(oracle):
    k = KAT["anchor_plane"]
    out = oracle.anchor_plane(k["height"], k["width"], k["stride"], k["base"])
    assert np.array_equal(out, np.array(k["out"], np.float32))
    assert np.array_equal(oracle.anchors_fpn(), np.array(KAT["anchors_fpn"]["out"], np.float32))


def test_decode_kats(oracle):
    k = KAT["bbox_pred"]
    got = oracle.bbox_pred(k["boxes"], k["deltas"])
    assert np.array_equal(got, np.array(k["out"], np.float32))
    k = KAT["landmark_pred"]
    got = oracle.landmark_pred(k["boxes"], k["deltas"]).reshape(2, 10)
    assert np.array_equal(got, np.array(k["out"], np.float32))
    k = KAT["clip_boxes"]
    b = np.array(k["boxes"], np.float32).reshape(-1, 4)
    got = oracle.clip_boxes(b, *k["im_shape"]).reshape(2, 8)
    assert np.array_equal(got, np.array(k["out"], np.float32))


def test_geometry_kats(oracle):
    for c in KAT["geometry"]["cases"]:
        nw, nh, sc = oracle.geometry(c["h"], c["w"])
        assert (nw, nh) == (c["out"][0], c["out"][1])
        assert sc == np.float32(c["out"][2])


def test_clip_nan_semantics(oracle):
    # Rust f32::min/max return the non-NaN operand: NaN.min(hi).max(0) == hi
    b = np.array([[np.nan, -5.0, 1e9, np.inf]], np.float32)
    assert oracle.clip_boxes(b, 640, 640).tolist() == [[639.0, 0.0, 639.0, 639.0]]


def test_argsort_is_stable(oracle):
    s = np.array([0.9, 0.8, 0.9, 0.7, 0.8, 0.9], np.float32)
    assert oracle.argsort_desc(s).tolist() == [0, 2, 5, 1, 4, 3]


def _rand_dets(rng, n, span=300.0):
    xy = rng.uniform(0, span, size=(n, 2))
    wh = rng.uniform(5, 120, size=(n, 2))
    sc = rng.uniform(0.1, 1.0, size=(n, 1))
    return np.concatenate([xy, xy + wh, sc], 1).astype(np.float32)


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 120), st.floats(0.05, 0.9))
def test_nms_properties(seed, n, thr):
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    d = _rand_dets(rng, n)
    keep = O.nms(d, thr)
    # keep is a subset, in non-increasing score order, starting with the global best
    assert len(set(keep.tolist())) == len(keep) and all(0 <= i < n for i in keep)
    assert np.all(np.diff(d[keep, 4]) <= 0)
    assert d[keep[0], 4] == d[:, 4].max()
    # idempotent: running NMS on the survivors keeps all of them
    assert O.nms(d[keep], thr).tolist() == list(range(len(keep)))
    # no two survivors overlap above the threshold; every suppressed box overlaps a better survivor
    def iou(a, b):
        w = max(0.0, np.float32(min(a[2], b[2]) - max(a[0], b[0]) + np.float32(1)))
        h = max(0.0, np.float32(min(a[3], b[3]) - max(a[1], b[1]) + np.float32(1)))
        i = np.float32(w) * np.float32(h)
        sa = (a[2] - a[0] + np.float32(1)) * (a[3] - a[1] + np.float32(1))
        sb = (b[2] - b[0] + np.float32(1)) * (b[3] - b[1] + np.float32(1))
        return i / (sa + sb - i)
    for x in range(len(keep)):
        for y in range(x + 1, len(keep)):
            assert iou(d[keep[x]], d[keep[y]]) <= np.float32(thr)
    # permuting the input rows does not change the kept SET when scores are distinct
    if len(np.unique(d[:, 4])) == n:
        p = rng.permutation(n)
        assert sorted(p[O.nms(d[p], thr)].tolist()) == sorted(keep.tolist())


def test_decode_nms_matches_stagewise_composition(oracle):
    """rfd_oracle_decode_nms == anchors -> bbox_pred -> clip -> threshold -> stable sort -> nms,
    composed in numpy from the single-stage oracle functions (face_detection.rs:319-470)."""
    H = W = 128
    heads = [h[0] for h in helpers.make_heads(7, 1, H, W, cand_rate=0.2, n_faces=3)]
    det, lmk, gidx, ncand = oracle.decode_nms(heads, H, W, 0.7, 0.45, det_scale=0.5)
    base = oracle.anchors_fpn()
    props, scores, lms, gids = [], [], [], []
    goff = 0
    for l, s in enumerate((32, 16, 8)):
        fh, fw = H // s, W // s
        plane = oracle.anchor_plane(fh, fw, s, base[l]).reshape(-1, 4)
        cls, bb, lm = heads[3 * l: 3 * l + 3]
        sc = cls[2:].transpose(1, 2, 0).reshape(-1)
        dl = bb.transpose(1, 2, 0).reshape(-1, 4)
        ld = lm.transpose(1, 2, 0).reshape(-1, 5, 2)
        boxes = oracle.clip_boxes(oracle.bbox_pred(plane, dl), H, W)
        lp = oracle.landmark_pred(plane, ld)
        sel = np.nonzero(sc >= np.float32(0.7))[0]
        props.append(boxes[sel]); scores.append(sc[sel]); lms.append(lp[sel]); gids.append(sel + goff)
        goff += fh * fw * 2
    props, scores, lms, gids = map(np.concatenate, (props, scores, lms, gids))
    assert ncand == len(scores)
    order = oracle.argsort_desc(scores)
    pre = np.concatenate([props[order], scores[order, None]], 1)
    keep = oracle.nms(pre, 0.45)
    assert np.array_equal(gidx, gids[order][keep])
    assert np.array_equal(det[:, 4], pre[keep, 4])
    assert np.array_equal(det[:, :4], pre[keep, :4] / np.float32(0.5))
    assert np.array_equal(lmk, lms[order][keep] / np.float32(0.5))


def test_golden_heads_fixture(oracle):
    """The committed seeded fixture (tests/golden/make_golden.py) is reproduced bit for bit."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "heads_128.npz"))
    heads = [z["h%d" % i] for i in range(9)]
    det, lmk, gidx, _ = oracle.decode_nms(heads, 128, 128, 0.7, 0.45, det_scale=float(z["det_scale"]))
    assert np.array_equal(det, z["det"]) and np.array_equal(lmk, z["lmk"]) and np.array_equal(gidx, z["gidx"])


def test_empty_and_special_values(oracle):
    H = W = 64
    heads = [h[0] for h in helpers.make_heads(3, 1, H, W, cand_rate=0.0)]
    det, lmk, gidx, ncand = oracle.decode_nms(heads, H, W)
    assert det.shape == (0, 5) and lmk.shape == (0, 5, 2) and ncand == 0
    # NaN scores fail `>= thr`; huge dw overflows exp to +inf and is clipped to the frame
    heads = [h[0].copy() for h in helpers.make_heads(4, 1, H, W, cand_rate=0.5)]
    heads[0][2:] = np.nan
    heads[4][2] = 100.0
    det, lmk, gidx, ncand = oracle.decode_nms(heads, H, W)
    assert np.all(gidx >= 8) and np.all(np.isfinite(det))
    assert np.all(det[:, :4] >= 0) and np.all(det[:, :4] <= 63)


def test_resize_identity_area_and_bounds(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(img, 37, 53), img)          # scale 1: identity
    big = rng.integers(0, 256, size=(40, 60, 3), dtype=np.uint8)
    half = oracle.resize_linear(big, 20, 30)                               # exact 2x: 2x2 mean
    ref = (big.astype(np.int32).reshape(20, 2, 30, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    assert np.array_equal(half, ref.astype(np.uint8))
    flat = np.full((33, 47, 3), 171, np.uint8)
    assert np.all(oracle.resize_linear(flat, 91, 17) == 171)               # constant stays constant
    up = oracle.resize_linear(img, 111, 160)
    assert up.min() >= img.min() and up.max() <= img.max()                 # convex combination


def test_preprocess_layout(oracle):
    img = helpers.make_image(5, 90, 160)
    det_img, tensor, sc = oracle.preprocess(img, 64, 64)
    nw, nh, s2 = oracle.geometry(90, 160, 64, 64)
    assert (nw, nh) == (64, 36) and sc == s2
    assert np.all(det_img[nh:] == 0)                                       # zero canvas below the paste
    assert np.array_equal(det_img[:nh, :nw], oracle.resize_linear(img, nh, nw))
    assert np.array_equal(tensor, det_img[..., ::-1].transpose(2, 0, 1).astype(np.float32))  # BGR->RGB planes


def test_face_selection_semantics(oracle):
    """FaceSelection::call (face_selection.rs:72-189) on hand-checked cases, image 720x1280."""
    k = np.arange(40, dtype=np.float32).reshape(4, 5, 2)
    b = np.array([[100, 100, 300, 350, .9],      # big but off-centre (centre x = 200 < 640 - 384)
                  [400, 200, 520, 340, .8],      # centred
                  [410, 210, 500, 300, .75],     # centred, smaller
                  [10, 10, 30, 30, .7]], np.float32)
    ob, ok = oracle.face_selection(b, k, 720, 1280)
    assert ob.tolist() == b[1].tolist() and np.array_equal(ok, k[1])
    ob, ok = oracle.face_selection(b, k, 720, 1280, is_enroll=True)   # enroll: biggest area wins
    assert ob.tolist() == b[0].tolist() and np.array_equal(ok, k[0])
    assert oracle.face_selection(b[:0], k[:0], 720, 1280) == (None, None)
    # no valid box (all tiny): falls back to ALL boxes, largest w+h, first maximum on ties
    t = np.array([[0, 0, 5, 5, .9], [600, 300, 605, 305, .8]], np.float32)
    ob, ok = oracle.face_selection(t, k[:2], 720, 1280)
    assert ob.tolist() == t[0].tolist()
    # key points come from the FIRST row within 2 px of the chosen box (:163-173)
    d = np.array([[401, 201, 521, 341, .9], [400, 200, 522, 342, .8]], np.float32)
    ob, ok = oracle.face_selection(d, k[:2], 720, 1280)
    assert ob.tolist() == d[1].tolist() and np.array_equal(ok, k[0])


def test_alignment_similarity_and_warp(oracle):
    """FaceAlignment restatement (face_alignment.rs:27-141): LMedS similarity + cv::warpAffine fixed point."""
    O = oracle
    # exact recovery of a known similarity
    th = np.deg2rad(-17.0)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    src = (O.STANDARD_LANDMARKS.astype(np.float64) @ R.T) * 1.7 + np.array([211.0, 95.0])
    M = O.estimate_similarity(src, O.STANDARD_LANDMARKS)
    assert np.abs(src.astype(np.float32) @ M[:, :2].T + M[:, 2] - O.STANDARD_LANDMARKS).max() < 1e-4
    assert abs(M[0, 0] - M[1, 1]) < 1e-12 and abs(M[0, 1] + M[1, 0]) < 1e-12          # 4 degrees of freedom
    assert O.estimate_similarity(np.tile([[5.0, 6.0]], (5, 1)), O.STANDARD_LANDMARKS) is None  # degenerate
    img = helpers.make_image(3, 200, 260, n_blobs=4)
    # identity map copies pixels; integer shifts too (weights 32767 + 1 at exact positions)
    eye = np.array([[1, 0, 0], [0, 1, 0]], np.float64)
    assert np.array_equal(O.warp_affine(img, eye, 112, 112), img[:112, :112])
    sh = np.array([[1, 0, -7], [0, 1, -9]], np.float64)
    assert np.array_equal(O.warp_affine(img, sh, 112, 112), img[9:121, 7:119])
    # fractional shift: within half a level of float bilinear
    T = np.array([[1, 0, -5.5], [0, 1, -3.25]], np.float64)
    w = O.warp_affine(img, T, 64, 64).astype(np.float64)
    f = img.astype(np.float64)
    e = 0.5 * (0.75 * f[3:67, 5:69] + 0.25 * f[4:68, 5:69]) + 0.5 * (0.75 * f[3:67, 6:70] + 0.25 * f[4:68, 6:70])
    assert np.abs(w - e).max() <= 0.5 + 1e-9
    # outside the frame: BORDER_CONSTANT 0, and the half-covered edge column blends with 0
    out = O.warp_affine(img, np.array([[1, 0, 40.5], [0, 1, 0]], np.float64), 8, 48)
    assert not out[:, :40].any()
    assert np.array_equal(out[:, 40], (img[:8, 0].astype(np.int32) * 16384 + 16384) >> 15)


def test_alignment_lmeds_restatement(oracle):
    """cv::estimateAffinePartial2D(LMEDS, 3.0, 2000, 0.99, 10) as face_alignment.rs:48-60 calls it, restated from OpenCV 4.x
    ptsetreg.cpp (parity unpinned: no OpenCV here).  Pins the restatement's own structure: the iteration count formula, the sample
    list of the re-seeded cv::RNG, outlier rejection, the inlier rule, and what it reduces to when every point is an inlier."""
    O = oracle
    assert O.cv_ransac_num_iters(0.99, 0.45, 2, 2000) == 13              # what the device kernel hard-codes (kLmedsIters)
    pairs = O.lmeds_samples(5)
    assert pairs.shape == (13, 2) and (pairs[:, 0] != pairs[:, 1]).all() and pairs.min() >= 0 and pairs.max() <= 4
    # regression pin of the restated multiply-with-carry sequence (state 2^64 - 1, coefficient 4164903690, draws modulo 5)
    assert pairs.tolist() == [[0, 4], [0, 3], [1, 2], [1, 0], [3, 0], [0, 4], [1, 4], [3, 1], [0, 3], [0, 1], [0, 4], [4, 1], [0, 3]]
    rng = np.random.default_rng(5)
    tmpl = O.STANDARD_LANDMARKS
    for trial in range(50):
        th, sc = rng.uniform(-0.6, 0.6), rng.uniform(0.4, 3.0)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        Minv = np.hstack([sc * R, rng.uniform(50, 400, (2, 1))])          # template -> image
        clean = (tmpl.astype(np.float64) @ Minv[:, :2].T + Minv[:, 2]).astype(np.float32)
        M_clean, inl = O.estimate_similarity(clean, tmpl, return_inliers=True)
        assert np.abs(clean @ M_clean[:, :2].T + M_clean[:, 2] - tmpl).max() < 2e-4
        # one landmark far off: every sampled pair that avoids it reproduces the other four, so it is rejected whichever it is
        k = trial % 5
        bad = clean.copy()
        bad[k] += np.float32(sc) * np.array([14.0, -9.0], np.float32)
        M_bad, inl = O.estimate_similarity(bad, tmpl, return_inliers=True)
        assert not inl[k] and inl.sum() == 4, (trial, inl)
        assert np.abs(M_bad - M_clean).max() < 1e-3
        M_all = O.estimate_similarity(bad, tmpl, all_points=True)       # rounds 1-3: dragged by the outlier
        assert np.abs(M_all - M_clean).max() > 1e-2
        # noisy but consistent points: the model is the least squares over whatever the inlier rule keeps (>= 3 of 5 here)
        noisy = clean + rng.normal(0, 0.4 * sc, clean.shape).astype(np.float32)
        M_n, inl = O.estimate_similarity(noisy, tmpl, return_inliers=True)
        assert inl.sum() >= 2
        sub = O.estimate_similarity(noisy[inl], tmpl[inl], all_points=True)
        assert np.abs(M_n - sub).max() < 1e-9
        if inl.all():
            assert np.abs(M_n - O.estimate_similarity(noisy, tmpl, all_points=True)).max() < 1e-12
    # two points: the kernel alone; coincident points: no model
    M2 = O.estimate_similarity(clean[:2], tmpl[:2])
    assert np.abs(clean[:2] @ M2[:, :2].T + M2[:, 2] - tmpl[:2]).max() < 1e-3
    assert O.estimate_similarity(np.tile([[5.0, 6.0]], (5, 1)), tmpl) is None


def test_alignment_branches_and_golden(oracle):
    O = oracle
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "align_112.npz"))
    crop, st = O.face_alignment(g["img"], g["box"], g["kps"])
    assert st == int(g["status"]) == 0 and np.array_equal(crop, g["crop"])
    assert np.allclose(O.estimate_similarity(g["kps"], O.STANDARD_LANDMARKS), g["M"], rtol=0, atol=1e-12)
    img = g["img"]                                           # 240 x 320
    same = np.tile([[100.0, 100.0]], (5, 1))
    # empty-transformation branch (:62-110) with the reference's quirks: x1 = max(x2 + 22, W), y1 = max(y1 + 22, H)
    crop, st = O.face_alignment(img, [60, 50, 200, 150, 0.9], same)
    assert st == 1
    assert np.array_equal(crop, O.resize_linear(img[28:240, 38:320], 112, 112))
    # no box: the 1/16 margins (:65-69)
    big = helpers.make_image(5, 400, 480, n_blobs=4)
    crop2, st2 = O.face_alignment(big, None, same)
    assert st2 == 1 and np.array_equal(crop2, O.resize_linear(big[3:400, 8:480], 112, 112))
    assert O.face_alignment(img, None, same)[1] == -1         # W - W/16 + 22 > W for frames narrower than 352
    # x2 + 22 > W: the Rect leaves the image -> Mat::roi error
    assert O.face_alignment(img, [60, 50, 310, 150, 0.9], same)[1] == -1


def test_resize_and_warp_against_independent_float_bilinear(oracle):
    """The oracle restates OpenCV's FIXED-POINT bilinear paths from their published algorithm (no OpenCV here).  As an
    independent cross-check, torch's float bilinear resampling of the same geometry (half-pixel centres, edge
    clamping / zero padding) must agree to within one grey level everywhere and ~0.25 on average."""
    import torch
    import torch.nn.functional as F
    O = oracle
    img = helpers.make_image(21, 300, 420, n_blobs=8)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
    for dh, dw in ((212, 297), (480, 672), (150, 210), (97, 420)):
        got = O.resize_linear(img, dh, dw).astype(np.float64)
        ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        if (dh, dw) == (150, 210):          # exact 2x: OpenCV's INTER_LINEAR switches to the 2x2 mean; so does bilinear at 2x
            assert np.abs(got - ref).max() <= 0.5 + 1e-6
        assert np.abs(got - ref).max() <= 1.0 + 1e-6 and np.abs(got - ref).mean() < 0.3, (dh, dw)
    # warpAffine vs grid_sample (zeros padding): rotation + scale + shift, partly outside the frame
    th = np.deg2rad(23.0)
    M = np.array([[0.6 * np.cos(th), -0.6 * np.sin(th), 31.5], [0.6 * np.sin(th), 0.6 * np.cos(th), -18.25]])
    got = O.warp_affine(img, M, 112, 112).astype(np.float64)
    A = np.vstack([M, [0, 0, 1]])
    Ai = np.linalg.inv(A)
    ys, xs = np.mgrid[0:112, 0:112].astype(np.float64)
    sx = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2]
    sy = Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
    H, W = img.shape[:2]
    grid = torch.from_numpy(np.stack([(2 * sx + 1) / W - 1, (2 * sy + 1) / H - 1], -1))[None].float()
    ref = F.grid_sample(t, grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0].permute(1, 2, 0).numpy()
    d = np.abs(got - ref)
    # 5-bit sub-pixel positions (1/32 px) on noise-like content: a few levels at most (measured 6.3), 0.8 on average
    assert d.max() <= 8.0 and d.mean() < 1.0
    # on a smooth image the same warp agrees to half a level (99.5th percentile) -- position quantisation, not a bias
    yy, xx = np.mgrid[0:300, 0:420]
    sm = np.stack([(xx * 0.5) % 256, (yy * 0.7) % 256, ((xx + yy) * 0.3) % 256], -1).astype(np.uint8)
    g2 = O.warp_affine(sm, M, 112, 112).astype(np.float64)
    r2 = F.grid_sample(torch.from_numpy(sm).permute(2, 0, 1)[None].float(), grid, mode="bilinear", padding_mode="zeros",
                       align_corners=False)[0].permute(1, 2, 0).numpy()
    d2 = np.abs(g2 - r2)
    assert np.percentile(d2, 99.5) <= 0.75 and d2.mean() < 0.2
    inside = (sx > 1) & (sx < W - 2) & (sy > 1) & (sy < H - 2)
    assert inside.sum() > 1000 and (~inside).sum() > 100


def test_device_expf_restatement_equals_the_host_libm(oracle):
    """The decode kernel computes f32::exp (face_detection.rs:534-535) by restating glibc's expf (kernels_post.hip: exp_cr).  The
    same operation sequence lives in the oracle library as rfd_oracle_expf_restated; here it is pinned against the host libm's
    expf -- what the oracle (and Rust's f32::exp) calls -- bit for bit: ~4.9 M inputs (every 512th f32 of +-[1e-6, 89], a dense sweep
    of the box-delta range +-[0, 6], specials).  Scanned offline over 2.2e8 inputs without a difference on a glibc 2.35 / FMA host;
    on a host whose glibc selects the non-FMA build of expf about 3 inputs in 10^9 differ by one ulp, hence the bar of 1."""
    lo, hi = np.float32(1e-6).view(np.uint32), np.float32(89.0).view(np.uint32)
    pos = np.arange(int(lo), int(hi), 512, dtype=np.uint32).view(np.float32)
    dense = np.linspace(-6, 6, 4_000_001, dtype=np.float64).astype(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 88.7, 88.8, 89.0, -103.9, -104.0, -87.5, 1e-40, -1e-40, float.fromhex('0x1.62e42ep6'), -float.fromhex('0x1.9fe368p6')], np.float32)
    x = np.concatenate([pos, -pos, dense, special])
    a, b = oracle.expf(x, restated=True), oracle.expf(x)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert x.size > 4_500_000 and int((~same).sum()) <= 1, "restated expf differs from the host libm at %s" % x[~same][:8]
    # and it is NOT the correctly rounded value everywhere: the reason the restatement exists
    cr = np.exp(dense.astype(np.float64)).astype(np.float32)
    assert 1e-6 < np.mean(cr != oracle.expf(dense)) < 1e-2


def test_device_expf_restatement_equals_the_recorded_libm_vectors(oracle):
    """The same restatement against committed vectors (tests/golden/expf_libm_glibc235.npz: 8 192 inputs and the outputs of glibc
    2.35's expf on an x86-64 FMA host, written by tests/golden/make_expf_golden.py), so that it is pinned independently of the
    libm of whatever host runs the tests."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "expf_libm_glibc235.npz"))
    got = oracle.expf(z["x"], restated=True)
    assert np.array_equal(got.view(np.uint32), z["y"].view(np.uint32))

"""FaceAlignment on the device (face_alignment.rs:27-141) against the CPU oracle.
Byte work -> bit-exact.  Parity against real OpenCV is unpinned (no OpenCV here; the reference's own test is
commented out); the similarity estimate restates OpenCV's LMedS (oracle/rfd_oracle.c) on both sides."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det(rfd):
    d = rfd.RetinaFaceDetection(max_batch_size=8, max_det=256)
    yield d
    d.close()


def test_golden_fixture(det, oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "align_112.npz"))
    crops, status = det.align_faces([g["img"]], [(g["box"], g["kps"])])
    assert status[0] == 0 and np.array_equal(crops[0], g["crop"])


def test_random_similarities_match_oracle(det, oracle):
    sizes = [(480, 640), (1080, 1920), (300, 200), (240, 320), (720, 1280), (97, 131), (640, 640), (2160, 3840)]
    frames = [helpers.make_image(100 + i, h, w, n_blobs=6) for i, (h, w) in enumerate(sizes)]
    for rnd in range(3):
        sel = []
        for i, (h, w) in enumerate(sizes):
            kps, box = helpers.make_face_kps(1000 * rnd + i, h, w)
            if (i + rnd) % 4 == 3:   # push the face over the frame edge: BORDER_CONSTANT taps
                kps = kps - np.array([0.45 * w, 0.3 * h], np.float32)
            sel.append((box, kps))
        crops, status = det.align_faces(frames, sel)
        for i, (f, (box, kps)) in enumerate(zip(frames, sel)):
            want, st = oracle.face_alignment(f, box, kps)
            assert status[i] == st == 0
            assert np.array_equal(crops[i], want), (rnd, i)


def test_outlier_landmarks_are_rejected_as_lmeds_does(det, oracle):
    """estimate_affine_partial_2d(LMEDS) (face_alignment.rs:48-60): a landmark that disagrees with the other four is left out of
    the model.  Device and oracle restate the same sample list, error arithmetic and inlier rule: byte-identical crops; and the
    model is the least squares over exactly the points the oracle reports as inliers, never the displaced one."""
    sizes = [(480, 640), (720, 1280), (1080, 1920), (300, 400), (640, 640)]
    frames = [helpers.make_image(300 + i, h, w, n_blobs=6) for i, (h, w) in enumerate(sizes)]
    for k in range(5):
        sel = []
        for i, (h, w) in enumerate(sizes):
            kps, box = helpers.make_face_kps(500 + 7 * k + i, h, w)
            kps[(k + i) % 5] += np.array([0.11 * w, -0.07 * h], np.float32)
            sel.append((box, kps))
        crops, status = det.align_faces(frames, sel)
        for i, (f, (box, kps)) in enumerate(zip(frames, sel)):
            want, st = oracle.face_alignment(f, box, kps)
            assert status[i] == st == 0 and np.array_equal(crops[i], want), (k, i)
            M, inl = oracle.estimate_similarity(kps, oracle.STANDARD_LANDMARKS, return_inliers=True)
            assert not inl[(k + i) % 5] and inl.sum() >= 2
            sub = oracle.estimate_similarity(kps[inl], oracle.STANDARD_LANDMARKS[inl], all_points=True)
            assert np.abs(M - sub).max() < 1e-9
            dragged = oracle.estimate_similarity(kps, oracle.STANDARD_LANDMARKS, all_points=True)   # rounds 1-3
            assert np.abs(M - dragged).max() > 1e-3


def test_other_template_sizes(det, oracle):
    f = helpers.make_image(7, 400, 500, n_blobs=5)
    kps, box = helpers.make_face_kps(3, 400, 500)
    tmpl = oracle.STANDARD_LANDMARKS * np.float32(224.0 / 112.0)
    crops, status = det.align_faces([f], [(box, kps)], image_size=(224, 200), standard_landmarks=tmpl)
    want, st = oracle.face_alignment(f, box, kps, image_size=(224, 200), standard_landmarks=tmpl)
    assert status[0] == st == 0 and crops.shape == (1, 200, 224, 3) and np.array_equal(crops[0], want)


def test_fallback_and_error_branches(det, oracle):
    f = helpers.make_image(9, 400, 480, n_blobs=5)
    same = np.tile(np.array([[100.0, 100.0]], np.float32), (5, 1))
    kps, box = helpers.make_face_kps(4, 400, 480)
    cases = [
        (np.array([60, 50, 200, 150, 0.9], np.float32), same),    # degenerate key points -> crop + resize (status 1)
        (np.array([60, 50, 470, 150, 0.9], np.float32), same),    # its Rect leaves the frame -> -3
        (box, None),                                              # no key points: the reference's call errors -> -1
        (None, None),                                             # nothing selected -> -2
        (box, kps),                                               # ordinary warp next to the special cases
        (np.array([206, 100, 310, 190, 0.9], np.float32), same),  # ROI 184..480 x 78..400: exact 2x? no: generic resize
    ]
    crops, status = det.align_faces([f] * len(cases), cases)
    assert status.tolist() == [1, -3, -1, -2, 0, 1]
    for i in (0, 4, 5):
        want, st = oracle.face_alignment(f, cases[i][0], cases[i][1])
        assert st == status[i] and np.array_equal(crops[i], want), i
    assert oracle.face_alignment(f, cases[1][0], same)[1] == -1
    for i in (1, 2, 3):
        assert not crops[i].any()
    # a ROI of exactly 224 x 224 takes OpenCV's 2x2-mean path (INTER_LINEAR -> INTER_AREA at scale 2)
    g = helpers.make_image(11, 224 + 30, 224 + 40, n_blobs=3)
    b2 = np.array([62, 52, 100, 100, 0.9], np.float32)         # x0 = 40, y0 = 30, x1 = W, y1 = H
    crops, status = det.align_faces([g], [(b2, same)])
    want, st = oracle.face_alignment(g, b2, same)
    assert st == status[0] == 1 and np.array_equal(crops[0], want)
    assert np.array_equal(want[0, 0], (g[30:32, 40:42].astype(np.int32).sum((0, 1)) + 2) >> 2)


def test_fused_detect_select_align(rfd, oracle):
    d = rfd.RetinaFaceDetection(max_batch_size=4, max_det=512, confidence_threshold=0.3)
    d.init_synthetic_weights(1234)
    frames = [helpers.make_image(200 + i, 480 + 32 * i, 640, n_blobs=5) for i in range(4)]
    sel, crops, status = d.detect_select_align(frames)
    sel2 = d.detect_select(frames)
    assert any(b is not None for b, _ in sel)
    for i, ((b, k), (b2, k2)) in enumerate(zip(sel, sel2)):
        assert (b is None) == (b2 is None) and (k is None) == (k2 is None)
        if b is None:
            assert status[i] == -2
            continue
        assert np.array_equal(b, b2)
        if k is None:
            assert status[i] == -1
            continue
        want, st = oracle.face_alignment(frames[i], b, k)
        assert status[i] == st and np.array_equal(crops[i], want), i
    d.close()

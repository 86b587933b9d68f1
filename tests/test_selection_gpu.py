"""GPU parity tests of the FaceSelection device epilogue (SURVEY.md section 8 row f-1;
reference src/pipeline/module/face_selection.rs:72-189) against the CPU oracle: bit-identical rows."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _rand_dets(rng, k, h, w, dup=False):
    xy = rng.uniform(-20, [w, h], size=(k, 2))
    wh = rng.uniform(5, [w / 2, h / 2], size=(k, 2))
    sc = np.sort(rng.uniform(0.7, 1.0, size=(k, 1)), 0)[::-1]
    d = np.concatenate([xy, xy + wh, sc], 1).astype(np.float32)
    if dup and k > 3:  # near-duplicates within 2 px: the reference takes the FIRST one's key points
        d[k // 2, :4] = d[k - 1, :4] + rng.uniform(-1.5, 1.5, size=4).astype(np.float32)
    kps = rng.uniform(0, max(h, w), size=(k, 5, 2)).astype(np.float32)
    return d, kps


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return np.array_equal(a, b)


@pytest.mark.parametrize("enroll", [False, True])
def test_select_matches_oracle(rfd, oracle, enroll):
    det = rfd.RetinaFaceDetection(max_batch_size=16, max_det=256)
    rng = np.random.default_rng(11 + enroll)
    for rep in range(6):
        sizes = [(int(rng.integers(200, 2200)), int(rng.integers(200, 3900))) for _ in range(16)]
        ks = [0, 1, 2, 3, 7, 40, 256, 5, 9, 100, 1, 33, 64, 65, 2, 17]
        dets = [_rand_dets(rng, k, h, w, dup=(i % 3 == 0)) for i, (k, (h, w)) in enumerate(zip(ks, sizes))]
        if rep == 0:
            dets[4] = (np.zeros((7, 5), np.float32), dets[4][1])          # all-zero boxes: nothing is "bigger than 0"
        if rep == 1:
            dets[5][0][:, :4] *= 0.01                                      # tiny faces: the valid/centre pools are empty
        got = det.select_faces(dets, sizes, is_enroll=enroll)
        for (d, k), (h, w), (gb, gk) in zip(dets, sizes, got):
            ob, ok = oracle.face_selection(d, k, h, w, is_enroll=enroll)
            assert _same(gb, ob) and _same(gk, ok)
    det.close()


def test_custom_margins(rfd, oracle):
    det = rfd.RetinaFaceDetection(max_batch_size=4, max_det=64)
    rng = np.random.default_rng(5)
    sizes = [(720, 1280)] * 4
    dets = [_rand_dets(rng, 30, 720, 1280) for _ in range(4)]
    cfg = (0.1, 0.2, 0.02, 0.05)
    got = det.select_faces(dets, sizes, cfg=cfg)
    for (d, k), (gb, gk) in zip(dets, got):
        ob, ok = oracle.face_selection(d, k, 720, 1280, False, *cfg)
        assert _same(gb, ob) and _same(gk, ok)
    det.close()


def test_detect_select_equals_detect_then_oracle_select(rfd, oracle):
    """Fused entry point = FacePipeline::extract lines 198-208: rfd_detect_batch rows -> oracle selection."""
    det = rfd.RetinaFaceDetection(max_batch_size=2, max_det=512, confidence_threshold=0.3)
    det.init_synthetic_weights(1234)
    frames = [helpers.make_image(31, 720, 1000), helpers.make_image(32, 1080, 1920)]
    rows = det.call_batch(frames)
    assert all(len(r[0]) > 0 for r in rows)
    for enroll in (False, True):
        got = det.detect_select(frames, is_enroll=enroll)
        for f, (d, k), (gb, gk) in zip(frames, rows, got):
            ob, ok = oracle.face_selection(d, k, f.shape[0], f.shape[1], is_enroll=enroll)
            assert _same(gb, ob) and _same(gk, ok)
    det.close()

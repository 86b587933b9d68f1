"""GPU parity tests of everything after the network: the HIP decode -> sort -> NMS -> rescale path
(through the C ABI: rfd_decode_nms / rfd_nms_sorted / _nms) against the CPU oracle on the same
seeded head tensors.  Bar: identical kept-anchor index sequences (bit-exact integer work) and
box / landmark coordinates within 1e-4 (BASELINE.json north_star); scores bit-exact."""
import ctypes
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
ATOL = 1e-4


@pytest.fixture(scope="module")
def det640(rfd):
    d = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=4, max_det=16800)
    yield d
    d.close()


def _compare(oracle, det, heads, n, H, W, scales, conf=0.7, iou=0.45):
    got = det.decode_nms(heads, scales, want_gidx=True)
    exact = total = 0
    for b in range(n):
        hb = [h[b] for h in heads]
        odet, olmk, ogidx, ncand = oracle.decode_nms(hb, H, W, conf, iou, det_scale=float(scales[b]))
        gdet, glmk, ggidx = got[b]
        assert np.array_equal(ggidx, ogidx), "kept index sequence differs (image %d)" % b
        assert det.last_total[b] == len(ogidx)
        assert np.array_equal(gdet[:, 4], odet[:, 4])
        # bit-identical since round 4 (glibc's expf restated in the decode kernel); north_star's bar is 1e-4
        assert np.array_equal(gdet[:, :4], odet[:, :4]), "box coordinates differ (image %d): max %g" % (b, np.nanmax(np.abs(gdet[:, :4] - odet[:, :4])))
        assert np.array_equal(glmk, olmk, equal_nan=True), "landmarks differ (image %d)" % b
        exact += int(np.sum(gdet == odet)) + int(np.sum((glmk == olmk) | (np.isnan(glmk) & np.isnan(olmk))))
        total += gdet.size + glmk.size
    return exact, total


def test_typical(rfd, oracle, det640):
    heads = helpers.make_heads(11, 4, cand_rate=0.006, n_faces=12)
    sc = np.array([1.0, 1 / 3, 0.625, 0.99791664], np.float32)
    exact, total = _compare(oracle, det640, heads, 4, 640, 640, sc)
    assert exact == total  # in practice every coordinate is bit-identical, not merely within 1e-4
    assert det640.stats()["candidates"] > 100


def test_dense_crowd(rfd, oracle, det640):
    # >500 clusters and thousands of candidates per image (BASELINE.json configs[4])
    heads = helpers.make_heads(12, 2, cand_rate=0.2, n_faces=600)
    exact, total = _compare(oracle, det640, heads, 2, 640, 640, np.array([1 / 6, 1 / 6], np.float32))
    assert exact >= 0.9999 * total


def test_mixed_batch_dense_and_sparse_images(rfd, oracle):
    """One launch, eight images: dense crowds (split over kNmsChunks workgroups that hand kept boxes to each other through
    global memory), sparse images (done by the first workgroup alone, the others exit) and an empty one, twice in a row
    (the progress words of the first launch must not satisfy the second: they carry the launch's epoch)."""
    d = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=8, max_det=16800)
    dense = helpers.make_heads(31, 4, cand_rate=0.25, n_faces=500)
    sparse = helpers.make_heads(32, 3, cand_rate=0.01, n_faces=10)
    empty = helpers.make_heads(33, 1, cand_rate=0.0)
    order = [("d", 0), ("s", 0), ("d", 1), ("e", 0), ("d", 2), ("s", 1), ("s", 2), ("d", 3)]
    src = {"d": dense, "s": sparse, "e": empty}
    heads = [np.stack([src[k][l][i] for k, i in order]) for l in range(9)]
    sc = np.linspace(0.3, 1.0, 8).astype(np.float32)
    for _ in range(2):
        _compare(oracle, d, heads, 8, 640, 640, sc)
    # the reverse order: different images land on different chunk workgroups
    heads_r = [h[::-1].copy() for h in heads]
    _compare(oracle, d, heads_r, 8, 640, 640, sc)
    assert d.stats()["candidates"] > 4 * 2048
    d.close()


def test_config5_dense_crowd_at_its_real_batch(rfd, oracle):
    """BASELINE.json configs[4] at its stated size: 64 dense-crowd images (16 distinct ones, each ~12 k candidates and > 500
    planted faces, four times in shuffled slots) in ONE launch -- 256 chunk workgroups of nms_chunked_kernel on 256 CUs, every one
    of which spins on its predecessor's progress word: the occupancy at which the hand-over protocol is most exposed.  Identical
    kept-anchor sequences per image against the oracle (nms.rs:3-65 semantics), and the spin_fail word stayed clear (a set
    word makes rfd_decode_nms fail)."""
    d = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=64, max_det=16800)
    base = helpers.make_heads(41, 16, cand_rate=0.7, n_faces=600)
    perm = np.random.default_rng(42).permutation(64) % 16
    heads = [h[perm] for h in base]
    sc = np.full(64, 1 / 6, np.float32)           # 3840x2160 sources
    for rep in range(2):                          # twice: the second launch must not be satisfied by the first one's epochs
        got = d.decode_nms(heads, sc, want_gidx=True)
        ref = {}
        for b in range(64):
            i = int(perm[b])
            if i not in ref:
                ref[i] = oracle.decode_nms([h[i] for h in base], 640, 640, 0.7, 0.45, det_scale=1 / 6)
            odet, olmk, ogidx, ncand = ref[i]
            gdet, glmk, ggidx = got[b]
            assert ncand > 11000 and len(ogidx) > 500
            assert np.array_equal(ggidx, ogidx), "kept index sequence differs (slot %d = image %d)" % (b, i)
            assert np.array_equal(gdet[:, 4], odet[:, 4])
            # Bit-identical since round 4: the device's exp is glibc's expf restated operation by operation (kernels_post.hip
            # exp_cr; pinned against the host's libm on the CPU in test_oracle_cpu.py), and the rescale is the same true f32
            # division (face_detection.rs:473-493).  Rounds 1-3 computed (float)exp((double)x) -- correctly rounded, which
            # glibc's expf is not always -- and needed 6e-4 here: 1 coordinate of 14 752 was off by one ulp at 4K scale.
            assert np.array_equal(gdet[:, :4], odet[:, :4]), "box coordinates differ (slot %d): max %g" % (b, np.abs(gdet[:, :4] - odet[:, :4]).max())
            assert np.array_equal(glmk, olmk)
    assert d.stats()["candidates"] > 64 * 11000
    d.close()


def test_all_anchors_pass_at_batch_64(rfd, oracle):
    """The worst case (every one of the 16 800 anchors is a candidate) on all 64 slots at once; 4 distinct images."""
    d = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=64, max_det=16800)
    base = helpers.make_heads(43, 4, cand_rate=1.0)
    heads = [np.concatenate([h] * 16) for h in base]
    got = d.decode_nms(heads, np.ones(64, np.float32), want_gidx=True)
    ref = [oracle.decode_nms([h[i] for h in base], 640, 640, 0.7, 0.45, det_scale=1.0) for i in range(4)]
    for b in range(64):
        odet, olmk, ogidx, ncand = ref[b % 4]
        assert ncand == 16800
        assert np.array_equal(got[b][2], ogidx)
        assert np.array_equal(got[b][0], odet)          # all 16 800 decodes of the image, bit for bit
    d.close()


def test_nms_give_up_flag_is_an_error_not_wrong_detections(rfd, oracle):
    """A chunk workgroup that times out waiting for its predecessor sets a device word; every call that synchronises reads it
    back and fails (the reference returns Err on every failure, face_detection.rs:498-509) -- then clears it."""
    d = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=2, max_det=2048)
    heads = helpers.make_heads(44, 2, cand_rate=0.2, n_faces=50)
    sc = np.ones(2, np.float32)
    ok = d.decode_nms(heads, sc)
    L = rfd.load_library()
    assert L.rfd_debug_poke_nms_flag(d._ctx, 1) == 0
    with pytest.raises(rfd.RfdError) as e:
        d.decode_nms(heads, sc)
    assert "NMS" in str(e.value)
    again = d.decode_nms(heads, sc)               # the word was cleared: the same call succeeds and gives the same result
    for (a, _), (b, _) in zip(ok, again):
        assert np.array_equal(a, b)
    assert e.value.status == rfd.RFD_ERR_HIP
    d.close()


def test_worst_case_all_anchors(rfd, oracle, det640):
    heads = helpers.make_heads(13, 1, cand_rate=1.0)
    _compare(oracle, det640, heads, 1, 640, 640, np.array([1.0], np.float32))
    assert det640.stats()["candidates"] == 16800


def test_ties_follow_stable_sort_order(rfd, oracle, det640):
    heads = helpers.make_heads(14, 2, cand_rate=0.05, n_faces=20, quantize=32)
    _compare(oracle, det640, heads, 2, 640, 640, np.array([1.0, 0.5], np.float32))


def test_empty(rfd, oracle, det640):
    heads = helpers.make_heads(15, 2, cand_rate=0.0)
    got = det640.decode_nms(heads, np.ones(2, np.float32))
    for d, k in got:
        assert d.shape == (0, 5) and k.shape == (0, 5, 2)   # face_detection.rs:413-419
    assert det640.last_total.tolist() == [0, 0]


def test_special_values(rfd, oracle, det640):
    heads = helpers.make_heads(16, 1, cand_rate=0.3)
    heads[0][0, 2:, :10] = np.nan          # NaN scores are filtered (`>=` is false)
    heads[4][0, 2] = 100.0                 # exp overflow -> inf -> clipped
    heads[4][0, 6] = -120.0                # exp underflow (subnormal / zero width)
    heads[7][0, 0, 5:9] = np.nan           # NaN deltas: clip maps NaN to the upper bound
    heads[8][0, 3, 2:4] = np.inf           # landmarks are not clipped
    _compare(oracle, det640, heads, 1, 640, 640, np.array([0.5], np.float32))


def test_thresholds_and_truncation(rfd, oracle):
    d = rfd.RetinaFaceDetection(image_size=(320, 256), max_batch_size=2, max_det=7,
                                confidence_threshold=0.5, iou_threshold=0.3)
    heads = helpers.make_heads(17, 2, 256, 320, cand_rate=0.1, n_faces=5)
    got = d.decode_nms(heads, np.ones(2, np.float32), want_gidx=True)
    for b in range(2):
        odet, olmk, ogidx, _ = oracle.decode_nms([h[b] for h in heads], 256, 320, 0.5, 0.3, 1.0)
        assert d.last_total[b] == len(ogidx) and len(ogidx) > 7
        assert np.array_equal(got[b][2], ogidx[:7])        # count = min(K, max_det), rows in kept order
        np.testing.assert_allclose(got[b][0], odet[:7], rtol=0, atol=ATOL)
    d.close()


def test_large_input_uses_streaming_nms_path(rfd, oracle):
    """1024x768 network input = 32256 anchors: beyond the register-resident NMS capacity (17408), so the
    LDS-cache / L2 streaming variant of the kernel runs."""
    d = rfd.RetinaFaceDetection(image_size=(1024, 768), max_batch_size=1, max_det=32256)
    heads = helpers.make_heads(21, 1, 768, 1024, cand_rate=0.3, n_faces=200)
    _compare(oracle, d, heads, 1, 768, 1024, np.array([0.4], np.float32))
    assert d.stats()["candidates"] > 8000
    d.close()


def test_golden_fixture(rfd):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "heads_128.npz"))
    d = rfd.RetinaFaceDetection(image_size=(128, 128), max_batch_size=1, max_det=400)
    heads = [z["h%d" % i][None] for i in range(9)]
    (gdet, glmk, ggidx), = d.decode_nms(heads, np.array([z["det_scale"]], np.float32), want_gidx=True)
    assert np.array_equal(ggidx, z["gidx"])
    np.testing.assert_allclose(gdet, z["det"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(glmk, z["lmk"], rtol=0, atol=ATOL)
    assert d.stats()["candidates"] == int(z["ncand"])
    d.close()


def test_nms_sorted_kat_and_random(rfd, oracle, det640):
    boxes = np.array([[100, 100, 210, 210, .72], [250, 250, 420, 420, .8], [220, 220, 320, 330, .92],
                      [100, 100, 210, 210, .6]], np.float32)          # src/processing/nms.rs:76-81
    order = oracle.argsort_desc(boxes[:, 4])
    keep = det640.nms_sorted(boxes[order], 0.4)
    assert order[keep].tolist() == [2, 1, 0]
    rng = np.random.default_rng(5)
    for n in (1, 63, 64, 65, 1000, 5000):
        xy = rng.uniform(0, 600, size=(n, 2)); wh = rng.uniform(4, 150, size=(n, 2))
        d = np.concatenate([xy, xy + wh, np.sort(rng.uniform(0, 1, size=(n, 1)), 0)[::-1]], 1).astype(np.float32)
        assert np.array_equal(det640.nms_sorted(d, 0.45), oracle.nms(d, 0.45))
    assert det640.nms_sorted(np.zeros((0, 5), np.float32), 0.45).size == 0


def test_reference_nms_symbol(rfd, oracle):
    """`_nms` keeps the reference's C signature (src/rcnn/gpu_nms.hpp:7)."""
    L = rfd.load_library()
    rng = np.random.default_rng(6)
    n = 300
    xy = rng.uniform(0, 300, size=(n, 2)); wh = rng.uniform(10, 90, size=(n, 2))
    d = np.concatenate([xy, xy + wh, np.sort(rng.uniform(0, 1, size=(n, 1)), 0)[::-1]], 1).astype(np.float32)
    keep = np.zeros(n, np.int32)
    num = ctypes.c_int(0)
    L._nms(keep.ctypes.data, ctypes.byref(num), d.ctypes.data, n, 5, 0.3, 0)
    assert np.array_equal(keep[:num.value], oracle.nms(d, 0.3))


def test_errors(rfd, det640):
    heads = helpers.make_heads(18, 1)
    with pytest.raises(rfd.RfdError) as e:
        det640.decode_nms([h[:, :, :10] for h in heads], np.ones(1, np.float32))
    assert e.value.status == rfd.RFD_ERR_INVALID_ARG
    big = helpers.make_heads(19, 5, cand_rate=0.0)
    with pytest.raises(rfd.RfdError) as e:
        det640.decode_nms(big, np.ones(5, np.float32))
    assert e.value.status == rfd.RFD_ERR_CAPACITY

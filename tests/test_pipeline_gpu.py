"""rfd_submit_batch / rfd_collect_batch: two batches in flight, H2D on its own stream.
Results are checked against the ORACLE (decode / sort / NMS / rescale of the head tensors of the same frames:
identical kept-anchor sequences, coordinates within 1e-4) and must equal the synchronous rfd_detect_batch on the same
frames, in submission order."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))


@pytest.mark.parametrize("pinned", [False, True])
def test_pipelined_equals_synchronous(rfd, pinned):
    n = 4
    det = rfd.RetinaFaceDetection(max_batch_size=n, max_det=512, confidence_threshold=0.3, backbone=rfd.BACKBONE_MNET025)
    det.init_synthetic_weights(1234)
    batches = []
    for b in range(5):
        frames = [helpers.make_image(300 + 10 * b + i, 360 + 40 * ((b + i) % 3), 480 + 32 * (i % 2), n_blobs=5)
                  for i in range(n if b != 3 else 2)]          # a ragged batch in the middle
        batches.append(frames)
    want = [det.call_batch(f) for f in batches]
    if sum(len(d) for w in want for d, _ in w) == 0:   # diagnostics for an intermittent failure seen twice in round 3
        g = rfd.Graph(rfd.BACKBONE_MNET025)
        w0, b0 = det.get_layer(0, g.layers[0])
        wl, bl = det.get_layer(len(g.layers) - 1, g.layers[-1])
        di, tn, sc = det.preprocess(batches[0])
        h = det.forward(tn)
        again = [det.call_batch(f) for f in batches]
        raise AssertionError("no detections: stats %s; |w0| %.4g |w_last| %.4g; input sum %.6g scales %s; head |mean| %s max fg %s; second try %s"
                             % (det.stats(), float(np.abs(w0).mean()), float(np.abs(wl).mean()), float(tn.sum()), sc,
                                [float(np.abs(x).mean()) for x in h[:9:3]], [float(x[:, 2:4].max()) for x in h[:9:3]],
                                [sum(len(d) for d, _ in w) for w in again]))
    if pinned:   # same frames, living in page-locked memory handed out by the library
        pin = []
        for frames in batches:
            pf = []
            for f in frames:
                buf = det.host_frames(1, f.shape[0], f.shape[1])[0]
                buf[...] = f
                pf.append(buf)
            pin.append(pf)
        batches = pin
    got = []
    det.submit(batches[0])
    for b in range(1, len(batches)):
        det.submit(batches[b])          # two in flight
        got.append(det.collect())
    got.append(det.collect())
    for g, w in zip(got, want):
        assert _same(g, w)
    # the synchronous entry still works afterwards
    assert _same(det.call_batch(batches[1]), want[1])
    det.close()


@pytest.mark.parametrize("backbone,n", [("r50", 8), ("mnet025", 4)])
def test_pipelined_matches_oracle(rfd, oracle, backbone, n):
    """collect() against the oracle's post-network path (face_detection.rs:319-493) on the device's head tensors of the
    same frames -- R50 at n = 8 runs the pass as two 4-image chains (the split path), MobileNet as one graph."""
    bb = rfd.BACKBONE_R50 if backbone == "r50" else rfd.BACKBONE_MNET025
    det = rfd.RetinaFaceDetection(max_batch_size=n, max_det=512, backbone=bb)
    det.init_synthetic_weights(1234)
    batches = [[helpers.make_image(4000 + 10 * b + i, 400 + 40 * ((b + i) % 3), 520 + 32 * (i % 2), n_blobs=5)
                for i in range(n if b != 2 else n - 1)] for b in range(4)]
    _, tn, _ = det.preprocess(batches[0])
    h = det.forward(tn)
    thr = float(np.quantile(np.concatenate([h[3 * l][:, 2:4].reshape(-1) for l in range(3)]), 0.99))
    det.set_thresholds(thr, 0.45)
    want = []
    for frames in batches:
        pre = [oracle.preprocess(f, 640, 640) for f in frames]
        heads = det.forward(np.stack([p[1] for p in pre]))
        want.append([oracle.decode_nms([x[b] for x in heads], 640, 640, np.float32(thr), 0.45, float(pre[b][2]))[:2]
                     for b in range(len(frames))])
    assert sum(len(d) for w in want for d, _ in w) > 20
    got = []
    det.submit(batches[0])
    for b in range(1, len(batches)):
        det.submit(batches[b])          # two in flight
        got.append(det.collect())
    got.append(det.collect())
    for bi, (g, w) in enumerate(zip(got, want)):
        assert len(g) == len(w)
        for i, ((gd, gk), (od, ok)) in enumerate(zip(g, w)):
            assert len(gd) == len(od), (bi, i)
            assert np.array_equal(gd[:, 4], od[:, 4]), (bi, i)
            np.testing.assert_allclose(gd[:, :4], od[:, :4], rtol=0, atol=1e-4)
            np.testing.assert_allclose(gk, ok, rtol=0, atol=1e-4)
    det.close()


def test_pipeline_state_errors(rfd):
    det = rfd.RetinaFaceDetection(max_batch_size=2, max_det=64, backbone=rfd.BACKBONE_MNET025)
    det.init_synthetic_weights(1)
    f = [helpers.make_image(1, 200, 300, n_blobs=2)]
    with pytest.raises(rfd.RfdError) as e:
        det.collect()                   # nothing in flight
    assert e.value.status == rfd.RFD_ERR_STATE
    det.submit(f)
    det.submit(f)
    with pytest.raises(rfd.RfdError) as e:
        det.submit(f)                   # a third batch
    assert e.value.status == rfd.RFD_ERR_STATE
    a, b = det.collect(), det.collect()
    assert _same(a, b)
    det.close()


def test_pipelined_many_detections_and_contiguous_frames(rfd):
    """Round 3's transfer paths: (i) images that keep more rows than the prefix rfd_submit_batch copies back unconditionally (128)
    have the rest fetched by rfd_collect_batch; (ii) frames lying back to back in one page-locked block travel as one copy.
    Both must give exactly the synchronous result."""
    n = 8
    det = rfd.RetinaFaceDetection(max_batch_size=n, max_det=2048, confidence_threshold=0.02, iou_threshold=0.9)
    det.init_synthetic_weights(1234)
    block = det.host_frames(n, 480, 640)            # one rfd_host_alloc block: frames contiguous, rows tight
    for i in range(n):
        block[i] = helpers.make_image(8100 + i, 480, 640, n_blobs=6)
    frames = [block[i] for i in range(n)]
    want = det.call_batch(frames)
    ks = [len(d) for d, _ in want]
    assert max(ks) > 128 and min(ks) >= 0, ks        # the low threshold / high IoU threshold keep hundreds of boxes
    det.submit(frames)
    det.submit(frames[:5])                           # a ragged second batch in flight
    got = det.collect()
    got2 = det.collect()
    assert _same(got, want) and _same(got2, want[:5])
    det.close()

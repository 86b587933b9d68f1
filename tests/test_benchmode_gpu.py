"""The configuration bench.py times -- RetinaFace-R50, batch 32, two 16-image chains on the tuned part streams, calls
overlapped across steps (rfd_detect_batch_device, async = 2) -- checked against the ORACLE, not against the library's
own synchronous result: byte-exact preprocess (face_detection.rs:131-232) and, on the head tensors of the same weights
and frames, identical kept-anchor sequences with coordinates within 1e-4 (north_star) after the oracle's decode / sort /
NMS / rescale (face_detection.rs:319-493, processing/nms.rs:3-65).  Parity status of the oracle itself: "parity
unpinned" (the reference holds no expected outputs; oracle/rfd_oracle.c header)."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu

B = 32
MAX_DET = 1024


def _oracle_rows(oracle, det, frames, thr):
    """(det, kps) per frame by the oracle's post-network path on the device's head tensors for these frames."""
    pre = [oracle.preprocess(f, 640, 640) for f in frames]
    di, tn, sc = det.preprocess(frames)
    for b, p in enumerate(pre):  # a2 / a3 at the bench's batch size: byte-exact
        assert np.array_equal(di[b], p[0]) and np.array_equal(tn[b], p[1]) and sc[b] == p[2], b
    heads = det.forward(tn)
    rows = []
    for b in range(len(frames)):
        odet, olmk, _, _ = oracle.decode_nms([h[b] for h in heads], 640, 640, np.float32(thr), 0.45, float(pre[b][2]))
        rows.append((odet, olmk))
    return rows, heads


def _check_slab(slab, want, tag):
    got = slab.unpack()
    tot = slab.total().cpu().numpy()
    for b, ((gd, gk), (od, ok)) in enumerate(zip(got, want)):
        assert len(gd) == len(od) == tot[b], (tag, b, len(gd), len(od), int(tot[b]))
        assert np.array_equal(gd[:, 4], od[:, 4]), (tag, b)                       # same anchors, same order
        np.testing.assert_allclose(gd[:, :4], od[:, :4], rtol=0, atol=1e-4, err_msg=str((tag, b)))
        np.testing.assert_allclose(gk, ok, rtol=0, atol=1e-4, err_msg=str((tag, b)))


@pytest.fixture(scope="module")
def det32(rfd):
    det = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=B, max_det=MAX_DET)
    det.init_synthetic_weights(1234)
    yield det
    det.close()


def test_benchmarked_mode_matches_oracle(rfd, oracle, det32):
    from rfd_hip import parallel
    det = det32
    dev = torch.device("cuda", 0)
    sets = []
    # two frame sets of mixed source sizes (letterboxed differently), alternating between consecutive calls
    for k in range(2):
        frames = [helpers.make_image(7000 + 100 * k + i, (640, 480, 720)[i % 3], (640, 640, 1000)[i % 3], n_blobs=5)
                  for i in range(B)]
        sets.append(frames)
    # score statistics of random weights are arbitrary: threshold at the 99.4 % quantile of the fg scores (~100 candidates / image)
    _, tn, _ = det.preprocess(sets[0][:4])
    h4 = det.forward(tn)
    fg = np.concatenate([h4[3 * l][:, 2:4].reshape(-1) for l in range(3)])
    thr = float(np.quantile(fg, 0.994))
    det.set_thresholds(thr, 0.45)
    want = [_oracle_rows(oracle, det, fr, thr)[0] for fr in sets]
    assert sum(len(d) for d, _ in want[0]) > 100 and sum(len(d) for d, _ in want[1]) > 100
    dev_sets = []
    for fr in sets:
        bufs = [torch.from_numpy(f).to(dev) for f in fr]
        dev_sets.append((bufs, [t.data_ptr() for t in bufs], [f.shape[:2] for f in fr]))
    slabs = [parallel.DetectionSlab(B, MAX_DET, device=dev) for _ in range(3)]
    stream = torch.cuda.current_stream()
    det.set_stream(stream.cuda_stream)            # as bench.py does
    det.detect_device(dev_sets[0][1], dev_sets[0][2], *slabs[0].pointers(), async_=0)   # set-up pass: stream tuning
    det.sync()
    _check_slab(slabs[0], want[0], "sync")
    for rnd in range(3):
        for s in slabs:
            s.buf.zero_()
        torch.cuda.synchronize()
        order = [(rnd + i) % 2 for i in range(3)]
        for i, k in enumerate(order):            # three calls in flight back to back, alternating frame sets
            det.detect_device(dev_sets[k][1], dev_sets[k][2], *slabs[i].pointers(), async_=2)
        det.sync()
        torch.cuda.synchronize()
        for i, k in enumerate(order):
            _check_slab(slabs[i], want[k], "round %d call %d set %d" % (rnd, i, k))
    det.set_stream(None)
    det.set_thresholds(0.7, 0.45)


def test_overlap_mode_with_changing_batch_sizes(rfd, oracle, det32):
    """The part boundary of the overlap mode is (n+1)/2: consecutive calls with different n use different workspace
    slices per chain, and other entry points run on the caller's stream.  Every call must still equal the oracle."""
    from rfd_hip import parallel
    det = det32
    dev = torch.device("cuda", 0)
    frames = [helpers.make_image(8100 + i, 640, 640, n_blobs=5) for i in range(B)]
    _, tn, _ = det.preprocess(frames[:4])
    h4 = det.forward(tn)
    thr = float(np.quantile(np.concatenate([h4[3 * l][:, 2:4].reshape(-1) for l in range(3)]), 0.994))
    det.set_thresholds(thr, 0.45)
    want, _ = _oracle_rows(oracle, det, frames, thr)
    bufs = [torch.from_numpy(f).to(dev) for f in frames]
    ptrs = [t.data_ptr() for t in bufs]
    seq = [32, 20, 9, 32, 16, 31, 8, 32]
    slabs = [parallel.DetectionSlab(n, MAX_DET, device=dev) for n in seq]
    for rep in range(2):
        for s in slabs:
            s.buf.zero_()
        torch.cuda.synchronize()
        for j, (n, s) in enumerate(zip(seq, slabs)):
            # frames [off, off+n): every call sees different images in its slices
            off = (5 * j) % (B - n + 1)
            det.detect_device(ptrs[off:off + n], [(640, 640)] * n, *s.pointers(), async_=2 if (j + rep) % 4 != 3 else 1)
        det.sync()
        for j, (n, s) in enumerate(zip(seq, slabs)):
            off = (5 * j) % (B - n + 1)
            _check_slab(s, want[off:off + n], "rep %d call %d n %d" % (rep, j, n))
    det.set_thresholds(0.7, 0.45)


def test_rccl_gather_behind_the_c_abi_world_size_1(rfd, det32):
    """rfd_comm_init / rfd_gather_detections / rfd_comm_destroy (include/rfd.h): the library's own RCCL all-gather of the
    detection slabs.  One GPU here, so world size 1: the gathered slabs must equal the local ones (layout + stream order);
    the N > 1 layout is covered by the gloo test of tests/test_parallel_cpu.py and has not run on hardware."""
    from rfd_hip import parallel
    det = det32
    dev = torch.device("cuda", 0)
    n = 8
    frames = [helpers.make_image(8800 + i, 640, 640, n_blobs=5) for i in range(n)]
    det.set_thresholds(0.3, 0.45)
    bufs = [torch.from_numpy(f).to(dev) for f in frames]
    slab = parallel.DetectionSlab(n, MAX_DET, device=dev)
    assert det.comm_info() == (0, 0)
    with pytest.raises(rfd.RfdError) as e:
        det.gather_detections(slab.pointers(), n, slab.pointers())
    assert e.value.status == rfd.RFD_ERR_STATE
    det.comm_init(rfd.RetinaFaceDetection.comm_unique_id(), 0, 1)
    assert det.comm_info() == (0, 1)
    with pytest.raises(rfd.RfdError) as e:
        det.comm_init(rfd.RetinaFaceDetection.comm_unique_id(), 0, 1)
    assert e.value.status == rfd.RFD_ERR_STATE
    gs = parallel.GatheredSlabs(1, n, MAX_DET, dev)
    for _ in range(3):
        gs.buf.zero_()
        det.detect_device([t.data_ptr() for t in bufs], [(640, 640)] * n, *slab.pointers(), async_=1)
        det.gather_detections(slab.pointers(), n, gs.pointers())   # stream-ordered behind NMS, no host sync in between
        det.sync()
        rows, tot = gs.unpack()
        want = slab.unpack()
        assert sum(len(d) for d, _ in want) > 0
        assert np.array_equal(tot, slab.total().cpu().numpy())
        for (gd, gk), (wd, wk) in zip(rows, want):
            assert np.array_equal(gd, wd) and np.array_equal(gk, wk)
    det.comm_destroy()
    assert det.comm_info() == (0, 0)
    det.set_thresholds(0.7, 0.45)


def test_config4_256_frames_of_1080p_as_eight_rank_slices(rfd, oracle, det32):
    """BASELINE.json configs[3] at its stated size on the one GPU there is: 256 synthetic 1920 x 1080 frames as eight 32-frame
    rank slices, one after the other through the production path of a rank -- frames resident in HBM, rfd_detect_batch_device
    with the calls overlapped across slices (async = 2), then the library's RCCL all-gather (rfd_gather_detections; world
    size 1, so a gather is a stream-ordered copy) INTO THAT RANK'S POSITION of the rank-major 256-frame slab an 8-GPU job
    assembles.  Every frame is then compared with the oracle (face_detection.rs:131-198 geometry + resize, :319-493 decode /
    NMS / rescale by det_scale = 1/3): preprocess byte-exact, identical kept sequences (bit-equal scores in order), coordinates
    within 1e-4 of the NETWORK-scale value (= 3e-4 after the true f32 division by 1/3; bit-identical in practice).
    What this cannot show is eight real ranks exchanging over xGMI: N > 1 has never run on hardware here."""
    from rfd_hip import parallel
    det = det32
    dev = torch.device("cuda", 0)
    WORLD, NL = 8, B
    base = [helpers.make_image(9900 + i, 1080, 1920, n_blobs=10) for i in range(8)]

    def frame(i):   # 256 distinct frames from 8 rendered ones: rolled by a per-frame offset, one channel inverted per group
        f = np.roll(base[i % 8], shift=(37 * (i // 8), 101 * (i // 8)), axis=(0, 1))
        if (i // 64) & 1:
            f = f.copy(); f[..., i % 3] = 255 - f[..., i % 3]
        return np.ascontiguousarray(f)

    _, tn, _ = det.preprocess([frame(i) for i in range(4)])
    h4 = det.forward(tn)
    thr = float(np.quantile(np.concatenate([h4[3 * l][:, 2:4].reshape(-1) for l in range(3)]), 0.994))
    det.set_thresholds(thr, 0.45)
    stream = torch.cuda.current_stream()
    det.set_stream(stream.cuda_stream)
    det.comm_init(rfd.RetinaFaceDetection.comm_unique_id(), 0, 1)
    gs = parallel.GatheredSlabs(WORLD, NL, MAX_DET, dev)          # the 256-frame rank-major slab
    gb, gl, gc, gt = gs.pointers()
    local = [parallel.DetectionSlab(NL, MAX_DET, device=dev) for _ in range(2)]   # a rank's own slab, double-buffered
    want, keep_alive = [], []
    try:
        for r in range(WORLD):
            frames = [frame(r * NL + i) for i in range(NL)]
            rows, _ = _oracle_rows(oracle, det, frames, thr)      # also asserts the byte-exact preprocess of all 32 frames
            want += rows
            bufs = [torch.from_numpy(f).to(dev) for f in frames]
            keep_alive.append(bufs)                               # the frames of a call in flight must stay allocated
            sl = local[r & 1]
            det.detect_device([t.data_ptr() for t in bufs], [(1080, 1920)] * NL, *sl.pointers(), async_=2)
            # rank r's position in the four rank-major arrays
            dst = (gb + 4 * r * NL * MAX_DET * 5, gl + 4 * r * NL * MAX_DET * 10, gc + 4 * r * NL, gt + 4 * r * NL)
            det.gather_detections(sl.pointers(), NL, dst)         # stream-ordered behind this slice's NMS
            if len(keep_alive) > 2:
                det.sync()
                keep_alive.pop(0)
        det.sync()
        torch.cuda.synchronize()
        got, tot = gs.unpack()
    finally:
        det.comm_destroy()
        det.set_stream(None)
        det.set_thresholds(0.7, 0.45)
    assert len(got) == len(want) == 256
    nkept = 0
    for i, ((gd, gk), (od, ok)) in enumerate(zip(got, want)):
        assert len(gd) == len(od) == tot[i], (i, len(gd), len(od), int(tot[i]))
        assert np.array_equal(gd[:, 4], od[:, 4]), i                                  # same anchors, same order
        np.testing.assert_allclose(gd[:, :4], od[:, :4], rtol=0, atol=3e-4, err_msg=str(i))
        np.testing.assert_allclose(gk, ok, rtol=0, atol=3e-4, err_msg=str(i))
        nkept += len(od)
    assert nkept > 256 * 8   # the letterboxed 1080p content fills 640 x 360 of the canvas: ~18 kept boxes per frame at this threshold
    exact = np.mean([np.array_equal(gd, od) and np.array_equal(gk, ok) for (gd, gk), (od, ok) in zip(got, want)])
    assert exact > 0.99   # in practice every frame is bit-identical

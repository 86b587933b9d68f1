"""The network pass may run (a) sequentially on one stream, (b) with the independent SSH/head chains on side
streams, (c) as 2-4 contiguous parts of the batch on their own streams over disjoint slices of the same workspace,
(d) replayed from a hipGraph (unsplit passes only).  Every structure must give bit-identical detections (a half-batch race on shared workspace buffers was
found and fixed with exactly this comparison)."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backbone", ["r50", "mnet025"])
def test_all_execution_structures_agree(rfd, backbone):
    bb = rfd.BACKBONE_R50 if backbone == "r50" else rfd.BACKBONE_MNET025
    n = 17  # odd: uneven parts
    det = rfd.RetinaFaceDetection(max_batch_size=n, max_det=2048, confidence_threshold=0.3, backbone=bb)
    det.init_synthetic_weights(1234)
    frames = [helpers.make_image(i, 320 + 16 * (i % 3), 480, n_blobs=3) for i in range(n)]

    def run():
        return [det.call_batch(frames) for _ in range(3)]  # 1st eager, 2nd captures a graph (if enabled), 3rd replays

    det.debug_set_concurrency(False, 8, 1, False)
    ref = run()[0]
    assert sum(len(d) for d, _ in ref) > 0
    for name, args in (("side streams", (True, 8, 1, False)), ("2 parts", (True, 8, 2, False)),
                       ("3 parts (6+6+5)", (True, 5, 3, True)), ("4 parts (5+4+4+4)", (True, 4, 4, False)),
                       ("graph", (False, 8, 1, True)), ("side streams + graph", (True, 8, 1, True))):
        det.debug_set_concurrency(*args)
        for it, got in enumerate(run()):
            for b, ((gd, gk), (rd, rk)) in enumerate(zip(got, ref)):
                assert np.array_equal(gd, rd) and np.array_equal(gk, rk), (name, it, b)
    det.close()


def _same(got, ref):
    return all(np.array_equal(a, c) and np.array_equal(b, d) for (a, b), (c, d) in zip(got, ref))


@pytest.mark.parametrize("victim,disturber", [("mnet025", "mnet025"), ("r50", "mnet025"), ("mnet025", "r50"), ("r50", "r50")])
def test_pipeline_is_exact_while_another_context_runs_convs(rfd, victim, disturber):
    """Regression for a cross-workgroup hazard seen on MI355X: with MFMA-issuing waves of ANOTHER kernel (a conv of another
    stream) on the same CU, broadcast ds_read_b128 reads of the old MobileNet first-conv weight table returned wrong
    data in lanes 48..63.  A second context keeps conv kernels in flight from another thread while the victim runs
    its whole pipeline; every result must stay bit-identical to the undisturbed one."""
    import threading
    import time
    bb = {"r50": rfd.BACKBONE_R50, "mnet025": rfd.BACKBONE_MNET025}
    n = 4
    frames = [helpers.make_image(40 + i, 640, 640, n_blobs=6) for i in range(n)]
    vic = rfd.RetinaFaceDetection(max_batch_size=n, max_det=2048, confidence_threshold=0.3, backbone=bb[victim])
    dis = rfd.RetinaFaceDetection(max_batch_size=n, max_det=2048, confidence_threshold=0.3, backbone=bb[disturber])
    for d in (vic, dis):
        d.init_synthetic_weights(1234)
    dis.debug_set_concurrency(False, 8, 1, False)
    dis.call_batch(frames)  # fills every tensor of the disturber, so single ops can be re-run
    convs = [k for k, o in enumerate(rfd.Graph(bb[disturber]).ops) if o.kind in (2, 6)]
    ref = vic.call_batch(frames)
    assert sum(len(d) for d, _ in ref) > 0
    stop = []

    def loop():
        j = 0
        while not stop:
            dis.debug_run(n, convs[j % len(convs)], convs[j % len(convs)])
            j += 1

    th = threading.Thread(target=loop)
    th.start()
    try:
        time.sleep(0.05)
        bad = sum(not _same(vic.call_batch(frames), ref) for _ in range(20))
    finally:
        stop.append(1)
        th.join()
    assert bad == 0
    vic.close()
    dis.close()


def test_independent_contexts_overlap_exactly(rfd):
    """Three contexts (own streams, own workspaces) launched back to back without waiting: each must reproduce its
    own sequential result."""
    import torch
    from rfd_hip import parallel
    n = 4
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(np.stack([helpers.make_image(60 + i, 640, 640, n_blobs=4) for i in range(n)])).to(dev)
    ptrs = [frames.data_ptr() + i * 640 * 640 * 3 for i in range(n)]
    ctxs = []
    for _ in range(3):
        det = rfd.RetinaFaceDetection(max_batch_size=n, max_det=512, confidence_threshold=0.3, backbone=rfd.BACKBONE_MNET025)
        det.init_synthetic_weights(1234)
        ctxs.append((det, parallel.DetectionSlab(n, 512, device=dev)))
    ref = []
    for det, slab in ctxs:
        det.detect_device(ptrs, [(640, 640)] * n, *slab.pointers(), async_=False)
        ref.append(slab.buf.clone())
    assert int(ctxs[0][1].count().sum()) > 0
    for _ in range(10):
        for det, slab in ctxs:
            det.detect_device(ptrs, [(640, 640)] * n, *slab.pointers(), async_=True)
        for det, _ in ctxs:
            det.sync()
        for (det, slab), r in zip(ctxs, ref):
            assert torch.equal(slab.buf, r)
    for det, _ in ctxs:
        det.close()


@pytest.mark.parametrize("backbone", ["r50", "mnet025"])
def test_cross_call_overlap_is_exact(rfd, backbone):
    """rfd_detect_batch_device(async = 2): the chains of call i+1 start while call i's tail / decode / NMS still run.
    Alternating frame sets and output slabs; every call must reproduce its synchronous result."""
    import torch
    from rfd_hip import parallel
    bb = rfd.BACKBONE_R50 if backbone == "r50" else rfd.BACKBONE_MNET025
    n = 9
    dev = torch.device("cuda", 0)
    det = rfd.RetinaFaceDetection(max_batch_size=n, max_det=512, confidence_threshold=0.3, backbone=bb)
    det.init_synthetic_weights(1234)
    sets, refs = [], []
    for k in range(3):
        fr = torch.from_numpy(np.stack([helpers.make_image(500 + 20 * k + i, 640, 640, n_blobs=4) for i in range(n)])).to(dev)
        sets.append((fr, [fr.data_ptr() + i * 640 * 640 * 3 for i in range(n)]))
    slabs = [parallel.DetectionSlab(n, 512, device=dev) for _ in range(3)]
    for (fr, ptrs), slab in zip(sets, slabs):
        det.detect_device(ptrs, [(640, 640)] * n, *slab.pointers(), async_=0)
        refs.append(slab.buf.clone())
    assert int(slabs[0].count().sum()) > 0 and not torch.equal(refs[0], refs[1])
    outs = [parallel.DetectionSlab(n, 512, device=dev) for _ in range(3)]
    for rnd in range(4):
        for o in outs:
            o.buf.zero_()
        torch.cuda.synchronize()
        for step in range(9):                      # 9 calls back to back, three of them in flight per output slab
            k = (step + rnd) % 3
            det.detect_device(sets[k][1], [(640, 640)] * n, *outs[k].pointers(), async_=2)
        det.sync()
        for k in range(3):
            assert torch.equal(outs[k].buf, refs[k]), (rnd, k)
        # mixing with the ordinary modes afterwards
        det.detect_device(sets[0][1], [(640, 640)] * n, *outs[1].pointers(), async_=1)
        det.sync()
        assert torch.equal(outs[1].buf, refs[0])
    det.close()


def test_fresh_contexts_always_have_their_weights(rfd):
    """Regression (round 3): the device fills of a new context ran on the NULL stream, unordered with the first weight uploads on
    the context's non-blocking stream; about one fresh context in 16 had layer 0's weights zeroed AFTER they were uploaded --
    every head constant, no detections, no error.  24 contexts back to back would have caught that 3 times out of 4."""
    g = rfd.Graph(rfd.BACKBONE_MNET025)
    frames = [helpers.make_image(77, 400, 520, n_blobs=5)]
    first = None
    for k in range(24):
        det = rfd.RetinaFaceDetection(max_batch_size=1, max_det=256, confidence_threshold=0.3, backbone=rfd.BACKBONE_MNET025)
        det.init_synthetic_weights(1234)
        w0, _ = det.get_layer(0, g.layers[0])
        assert float(np.abs(w0).mean()) > 0, "context %d lost the weights of layer 0" % k
        d, kps = det.call_batch(frames)[0]
        if first is None:
            first = d
            assert len(d) > 0
        assert np.array_equal(d, first), k
        det.close()

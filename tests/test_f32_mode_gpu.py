"""The f32 parity mode (rfd_config.precision = RFD_PRECISION_F32, csrc/kernels_f32.hip) against the torch-CPU walk of the SAME op
list without any bf16 rounding (tests/torch_ref.py, round_bf16=False): every fusion of the graph -- shortcut convs as second K
segment, BN+ReLU on the consumer's operand, SSH sibling convs with two destinations, nearest-2x residual after ReLU (FPN), the
stage-1 back-to-back pairs, the fused stem, softmax heads -- evaluated in f32 on the device and on the CPU from the weights the
device reports.  tests/test_t2_gpu.py holds the end-to-end leg against the unfused, BatchNorm-explicit model."""
import numpy as np
import pytest
import torch

import helpers
import torch_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,n", [((640, 640), 2), ((320, 256), 3)])
def test_f32_mode_heads_match_torch_walk_of_the_same_graph(rfd, oracle, size, n):
    w, h = size
    det = rfd.RetinaFaceDetection(image_size=size, max_batch_size=n, max_det=2048, precision=rfd.PRECISION_F32)
    det.init_synthetic_weights(4321)
    g = rfd.Graph(rfd.BACKBONE_R50, w, h)
    ref = torch_ref.TorchRef(g, det, round_bf16=False, acc64=True)   # get_layer returns the f32 values the f32 kernels use; f64-accumulating walk
    frames = [helpers.make_image(7700 + i, h + 37 * i, w - 50 * i, n_blobs=6) for i in range(n)]
    pre = [oracle.preprocess(f, w, h) for f in frames]
    tensor = np.stack([p[1] for p in pre])
    dev = det.forward(tensor)
    x4 = torch.cat([torch.from_numpy(tensor), torch.zeros(n, 1, h, w)], 1)
    want = ref.heads(ref.forward(x4))
    for k, (a, b) in enumerate(zip(dev, want)):
        rel = float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-12))
        assert rel < 2e-6, (k, rel)   # both sides sum in f64 and round once per layer (round 3, f32 sums on both sides: < 2e-5)
    # the whole path in f32 mode: preprocess byte-exact as always, then decode / NMS of these heads = the oracle's
    heads_np = [np.ascontiguousarray(x) for x in dev]
    fg = np.concatenate([heads_np[3 * l][:, 2:4].reshape(n, -1) for l in range(3)], 1)
    thr = float(np.quantile(fg, 0.995))
    det.set_thresholds(thr, 0.45)
    got = det.call_batch(frames)
    assert sum(len(d) for d, _ in got) > 0
    for b in range(n):
        odet, olmk, ogidx, _ = oracle.decode_nms([x[b] for x in heads_np], h, w, np.float32(thr), 0.45, float(pre[b][2]))
        assert len(odet) == len(got[b][0])
        assert np.array_equal(got[b][0][:, 4], odet[:, 4])
        np.testing.assert_allclose(got[b][0][:, :4], odet[:, :4], rtol=0, atol=1e-4)
        np.testing.assert_allclose(got[b][1], olmk, rtol=0, atol=1e-4)
    det.close()


def test_f32_mode_is_r50_only_and_validated(rfd):
    with pytest.raises(rfd.RfdError) as e:
        rfd.RetinaFaceDetection(precision=rfd.PRECISION_F32, backbone=rfd.BACKBONE_MNET025)
    assert e.value.status == rfd.RFD_ERR_INVALID_ARG
    with pytest.raises(rfd.RfdError) as e:
        rfd.RetinaFaceDetection(precision=7)
    assert e.value.status == rfd.RFD_ERR_INVALID_ARG


def test_f32_mode_weight_file_round_trip_and_guards(rfd, tmp_path):
    """RFDW save / load carries the unrounded f32 values in this mode; hooks that assume the bf16 layout say so."""
    det = rfd.RetinaFaceDetection(image_size=(320, 256), max_batch_size=1, precision=rfd.PRECISION_F32)
    det.init_synthetic_weights(99)
    frames = [helpers.make_image(31, 256, 320, n_blobs=4)]
    _, tn, _ = det.preprocess(frames)
    h0 = det.forward(tn)
    path = str(tmp_path / "w.rfdw")
    det.save_weights(path)
    det2 = rfd.RetinaFaceDetection(image_size=(320, 256), max_batch_size=1, precision=rfd.PRECISION_F32)
    det2.load_weights(path)
    h1 = det2.forward(tn)
    for a, b in zip(h0, h1):
        assert np.array_equal(a, b)
    g = rfd.Graph(rfd.BACKBONE_R50, 320, 256)
    w, _ = det.get_layer(5, g.layers[5])
    assert np.any(helpers.bf16_round(w) != w)          # unrounded values, not the bf16 copy
    with pytest.raises(rfd.RfdError) as e:
        det.set_profiling(True)
    assert e.value.status == rfd.RFD_ERR_STATE
    det.close(); det2.close()

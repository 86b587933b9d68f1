"""Unfolded parameters (conv + BatchNorm) imported with rfd_hip.convert reproduce, on the device, the f32 torch model
written directly from SURVEY.md Appendix B (tests/unfolded_ref.py): topology, FPN tap points, SSH wiring, head packing
and the BN folding convention in one check.  bf16 activations/weights on the device vs f32 in torch: the bars below are
statistical, the bit-level checks of the kernels live in test_network_gpu.py."""
import numpy as np
import pytest

import helpers
import unfolded_ref

pytestmark = pytest.mark.gpu


def test_imported_network_matches_the_unfolded_model(rfd, oracle):
    from rfd_hip import convert
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    det = rfd.RetinaFaceDetection(max_batch_size=2, max_det=64)
    P = unfolded_ref.make_params(20241004)
    used = convert.import_unfolded(det, g, P)
    assert used == set(P)
    frames = [helpers.make_image(900 + i, 640, 640, n_blobs=6) for i in range(2)]
    _, tensor, _ = det.preprocess(frames)
    got = det.forward(tensor)
    want = unfolded_ref.forward(P, __import__("torch").from_numpy(tensor))
    names = ["%s%d" % (k, st) for st in (32, 16, 8) for k in ("cls", "bbox", "lmk")]
    for nm, a, b in zip(names, got, want):
        assert a.shape == b.shape, nm
        err = np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-12)
        corr = np.corrcoef(a.ravel(), b.ravel())[0, 1]
        assert err < 0.05 and corr > 0.998, (nm, err, corr)   # bf16 noise through ~60 layers: ~1-2 %
        if nm.startswith("cls"):   # soft-max probabilities: a logit difference of the bf16 size moves a few near 0.5
            d = np.abs(a - b)
            assert d.mean() < 0.01 and np.percentile(d, 99.0) < 0.08 and d.max() < 0.3, (nm, d.mean(), np.percentile(d, 99.0), d.max())
        print(nm, "rel L2 %.4f corr %.5f" % (err, corr))
    # a deliberately wrong import (BN1 of a unit dropped) must NOT pass the same bar: the check has teeth
    P2 = dict(P)
    P2["stage3_unit4_bn1_gamma"] = np.ones_like(P["stage3_unit4_bn1_gamma"])
    P2["stage3_unit4_bn1_beta"] = np.zeros_like(P["stage3_unit4_bn1_beta"])
    convert.import_unfolded(det, g, P2)
    bad = det.forward(tensor)
    errs = [np.linalg.norm((a - b).ravel()) / np.linalg.norm(b.ravel()) for a, b in zip(bad, want)]
    assert max(errs) > 0.05
    det.close()

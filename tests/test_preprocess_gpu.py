"""GPU parity tests of the letterbox/resize/tensorise kernel (rfd_preprocess through the C ABI)
against the CPU oracle: byte-exact det_img, exact f32 tensor, exact det_scale."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

SIZES = [(1080, 1920), (768, 1024), (640, 480), (480, 641), (1280, 1280), (720, 1280), (100, 100),
         (479, 641), (33, 1000), (2160, 3840), (640, 640), (1, 1), (7, 5)]


@pytest.fixture(scope="module")
def det(rfd):
    d = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=4, max_det=16)
    yield d
    d.close()


@pytest.mark.parametrize("hw", SIZES)
def test_matches_oracle(rfd, oracle, det, hw):
    img = helpers.make_image(hw[0] * 7 + hw[1], *hw)
    di, tn, sc = det.preprocess([img])
    odi, otn, osc = oracle.preprocess(img, 640, 640)
    assert sc[0] == osc
    assert np.array_equal(di[0], odi)
    assert np.array_equal(tn[0], otn)


def test_batch_of_mixed_sizes_and_strided_rows(rfd, oracle, det):
    imgs = [helpers.make_image(1, 300, 500), helpers.make_image(2, 900, 700), helpers.make_image(3, 1280, 2560)]
    wide = helpers.make_image(4, 200, 400)
    imgs.append(wide[:, 37:337])          # a view with padded rows: stride 1200 bytes, width 300
    di, tn, sc = det.preprocess(imgs)
    for i, im in enumerate(imgs):
        odi, otn, osc = oracle.preprocess(np.ascontiguousarray(im), 640, 640)
        assert sc[i] == osc and np.array_equal(di[i], odi) and np.array_equal(tn[i], otn)


def test_non_square_network_input(rfd, oracle):
    d = rfd.RetinaFaceDetection(image_size=(320, 256), max_batch_size=1, max_det=16)
    img = helpers.make_image(9, 480, 640)
    di, tn, sc = d.preprocess([img])
    odi, otn, osc = oracle.preprocess(img, 320, 256)
    assert sc[0] == osc and np.array_equal(di[0], odi) and np.array_equal(tn[0], otn)
    d.close()


def test_rejects_bad_frames(rfd, det):
    with pytest.raises(rfd.RfdError):
        det.preprocess([np.zeros((10, 10), np.uint8)])           # 1-channel: SURVEY.md A.7
    with pytest.raises(rfd.RfdError):
        det.preprocess([np.zeros((10, 10, 3), np.float32)])
    with pytest.raises(rfd.RfdError) as e:
        det.preprocess([np.zeros((1, 5000, 3), np.uint8)])       # letterbox height truncates to 0
    assert e.value.status == rfd.RFD_ERR_INVALID_ARG

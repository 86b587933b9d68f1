"""ctypes binding of oracle/librfd_oracle.so (the CPU restatement in rfd_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  Parity status: "parity unpinned"
(see the header of rfd_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librfd_oracle.so")
_lib = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

STRIDES = (32, 16, 8)
NUM_ANCHORS = 2


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "rfd_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librfd_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.rfd_oracle_generate_anchors2.argtypes = [C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p]
        L.rfd_oracle_generate_anchors2.restype = C.c_int
        L.rfd_oracle_anchors_fpn.argtypes = [_f32p]
        L.rfd_oracle_anchor_plane.argtypes = [C.c_int, C.c_int, C.c_int, _f32p, C.c_int, _f32p]
        L.rfd_oracle_bbox_pred.argtypes = [_f32p, _f32p, C.c_int, _f32p]
        L.rfd_oracle_expf_restated.argtypes = [_f32p, C.c_int, _f32p]
        L.rfd_oracle_expf_libm.argtypes = [_f32p, C.c_int, _f32p]
        L.rfd_oracle_landmark_pred.argtypes = [_f32p, _f32p, C.c_int, _f32p]
        L.rfd_oracle_clip_boxes.argtypes = [_f32p, C.c_int, C.c_int, C.c_int]
        L.rfd_oracle_argsort_desc.argtypes = [_f32p, C.c_int, _i32p]
        L.rfd_oracle_nms.argtypes = [_f32p, C.c_int, C.c_float, _i32p]
        L.rfd_oracle_nms.restype = C.c_int
        L.rfd_oracle_decode_nms.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_float,
                                            C.c_float, C.c_float, _f32p, _f32p, _i32p, C.c_int,
                                            C.POINTER(C.c_int)]
        L.rfd_oracle_decode_nms.restype = C.c_int
        L.rfd_oracle_geometry.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                          C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.rfd_oracle_resize_linear_u8c3.argtypes = [_u8p, C.c_int, C.c_int, C.c_ssize_t, _u8p,
                                                    C.c_int, C.c_int, C.c_ssize_t]
        L.rfd_oracle_preprocess.argtypes = [_u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p]
        L.rfd_oracle_preprocess.restype = C.c_float
        L.rfd_oracle_face_selection.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                                C.c_float, C.c_float, C.c_int, _f32p, _f32p, C.POINTER(C.c_int)]
        L.rfd_oracle_face_selection.restype = C.c_int
        _f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        L.rfd_oracle_estimate_similarity.argtypes = [_f32p, _f32p, C.c_int, _f64p]
        L.rfd_oracle_estimate_similarity.restype = C.c_int
        L.rfd_oracle_warp_affine_u8c3.argtypes = [_u8p, C.c_int, C.c_int, C.c_ssize_t, _f64p, _u8p, C.c_int, C.c_int]
        L.rfd_oracle_warp_affine_u8c3.restype = None
        L.rfd_oracle_face_alignment.argtypes = [_u8p, C.c_int, C.c_int, C.c_ssize_t, C.c_void_p, C.c_void_p, _f32p, C.c_int,
                                                C.c_int, _u8p]
        L.rfd_oracle_face_alignment.restype = C.c_int
        _lib = L
    return _lib


def generate_anchors2(base_size, ratios, scales):
    r = np.ascontiguousarray(ratios, np.float32)
    s = np.ascontiguousarray(scales, np.float32)
    out = np.zeros((len(r) * len(s), 4), np.float32)
    lib().rfd_oracle_generate_anchors2(base_size, r, len(r), s, len(s), out)
    return out


def anchors_fpn():
    out = np.zeros((3, 2, 4), np.float32)
    lib().rfd_oracle_anchors_fpn(out)
    return out


def anchor_plane(height, width, stride, base):
    base = np.ascontiguousarray(base, np.float32)
    out = np.zeros((height, width, base.shape[0], 4), np.float32)
    lib().rfd_oracle_anchor_plane(height, width, stride, base, base.shape[0], out)
    return out


def expf(x, restated=False):
    """the host libm's expf (what Rust's f32::exp calls), or the device kernel's restatement of it evaluated on the CPU"""
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    out = np.empty_like(x)
    (lib().rfd_oracle_expf_restated if restated else lib().rfd_oracle_expf_libm)(x, x.size, out)
    return out


def bbox_pred(boxes, deltas):
    boxes = np.ascontiguousarray(boxes, np.float32)
    deltas = np.ascontiguousarray(deltas, np.float32)
    out = np.zeros_like(boxes)
    lib().rfd_oracle_bbox_pred(boxes, deltas, boxes.shape[0], out)
    return out


def landmark_pred(boxes, deltas):
    boxes = np.ascontiguousarray(boxes, np.float32)
    deltas = np.ascontiguousarray(deltas, np.float32).reshape(-1, 5, 2)
    out = np.zeros_like(deltas)
    lib().rfd_oracle_landmark_pred(boxes, deltas, boxes.shape[0], out)
    return out


def clip_boxes(boxes, im_h, im_w):
    b = np.array(boxes, np.float32, copy=True, order="C")
    lib().rfd_oracle_clip_boxes(b, b.shape[0], im_h, im_w)
    return b


def argsort_desc(scores):
    s = np.ascontiguousarray(scores, np.float32)
    o = np.zeros(s.shape[0], np.int32)
    lib().rfd_oracle_argsort_desc(s, s.shape[0], o)
    return o


def nms(dets, thresh):
    d = np.ascontiguousarray(dets, np.float32)
    keep = np.zeros(max(d.shape[0], 1), np.int32)
    k = lib().rfd_oracle_nms(d, d.shape[0], float(thresh), keep)
    return keep[:k].copy()


def head_shapes(net_h, net_w):
    """The 9 output tensors of the reference's Triton contract (batch dim dropped)."""
    shapes = []
    for s in STRIDES:
        h, w = net_h // s, net_w // s
        shapes += [(2 * NUM_ANCHORS, h, w), (4 * NUM_ANCHORS, h, w), (10 * NUM_ANCHORS, h, w)]
    return shapes


def decode_nms(heads, net_h, net_w, conf_thr=0.7, iou_thr=0.45, det_scale=1.0):
    """heads: 9 f32 arrays [C,h,w] in the order 32,16,8 x (cls,bbox,lmk).
    Returns det [K,5], lmk [K,5,2], gidx [K] (global anchor index), n_candidates."""
    hs = [np.ascontiguousarray(h, np.float32) for h in heads]
    for h, shp in zip(hs, head_shapes(net_h, net_w)):
        assert h.shape == shp, (h.shape, shp)
    ptrs = (C.c_void_p * 9)(*[h.ctypes.data for h in hs])
    cap = sum(s[1] * s[2] for s in head_shapes(net_h, net_w)[::3]) * NUM_ANCHORS
    det = np.zeros((cap, 5), np.float32)
    lmk = np.zeros((cap, 5, 2), np.float32)
    gidx = np.zeros(cap, np.int32)
    ncand = C.c_int(0)
    k = lib().rfd_oracle_decode_nms(ptrs, net_h, net_w, conf_thr, iou_thr, det_scale, det, lmk,
                                    gidx, cap, C.byref(ncand))
    assert k >= 0
    return det[:k].copy(), lmk[:k].copy(), gidx[:k].copy(), ncand.value


def geometry(img_h, img_w, size_w=640, size_h=640):
    nw, nh, sc = C.c_int(), C.c_int(), C.c_float()
    lib().rfd_oracle_geometry(img_h, img_w, size_w, size_h, C.byref(nw), C.byref(nh), C.byref(sc))
    return nw.value, nh.value, np.float32(sc.value)


def resize_linear(src, dh, dw):
    src = np.ascontiguousarray(src, np.uint8)
    assert src.ndim == 3 and src.shape[2] == 3
    dst = np.zeros((dh, dw, 3), np.uint8)
    lib().rfd_oracle_resize_linear_u8c3(src, src.shape[0], src.shape[1], src.strides[0], dst, dh,
                                        dw, dst.strides[0])
    return dst


def preprocess(src, size_w=640, size_h=640):
    """src: HxWx3 u8 (BGR). Returns det_img [size_h,size_w,3] u8, tensor [3,size_h,size_w] f32
    (R,G,B planes, raw 0..255), det_scale."""
    src = np.ascontiguousarray(src, np.uint8)
    det_img = np.zeros((size_h, size_w, 3), np.uint8)
    tensor = np.zeros((3, size_h, size_w), np.float32)
    sc = lib().rfd_oracle_preprocess(src, src.shape[0], src.shape[1], src.strides[0], size_w,
                                     size_h, det_img.ctypes.data, tensor.ctypes.data)
    return det_img, tensor, np.float32(sc)


def face_selection(boxes, kps, img_h, img_w, is_enroll=False, margin_center_left_ratio=0.3,
                   margin_center_right_ratio=0.3, margin_edge_ratio=0.1, minimum_face_ratio=0.0075):
    """FaceSelection::call (face_selection.rs:72-189), defaults = FaceSelectionConfig::new (config.rs:107-117).
    Returns (box [5] or None, kps [5,2] or None)."""
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 5)
    k = np.ascontiguousarray(kps, np.float32).reshape(-1, 10)
    ob = np.zeros(5, np.float32)
    ok = np.zeros(10, np.float32)
    found = C.c_int(0)
    r = lib().rfd_oracle_face_selection(b if len(b) else np.zeros((1, 5), np.float32),
                                        k if len(k) else np.zeros((1, 10), np.float32), b.shape[0], img_h, img_w,
                                        margin_center_left_ratio, margin_center_right_ratio, margin_edge_ratio,
                                        minimum_face_ratio, 1 if is_enroll else 0, ob, ok, C.byref(found))
    if not r:
        return None, None
    return ob, (ok.reshape(5, 2) if found.value else None)


# FaceAlignmentConfig::new (config.rs:44-56)
STANDARD_LANDMARKS = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366], [41.5493, 92.3655],
                               [70.7299, 92.2041]], np.float32)


def estimate_similarity(src, dst, all_points=False, return_inliers=False):
    """cv::estimateAffinePartial2D(src -> dst, LMEDS, 3.0, 2000, 0.99, 10) restated (face_alignment.rs:48-60): 13 fixed two-point
    samples, least median, inlier rule, least squares over the inliers.  all_points: the closed form over every point (rounds 1-3).
    Returns the 2x3 f64 matrix or None (no model: the reference's empty-matrix branch)."""
    s = np.ascontiguousarray(src, np.float32).reshape(-1, 2)
    d = np.ascontiguousarray(dst, np.float32).reshape(-1, 2)
    M = np.zeros(6, np.float64)
    L = lib()
    if all_points:
        L.rfd_oracle_estimate_similarity_all_points.argtypes = L.rfd_oracle_estimate_similarity.argtypes
        L.rfd_oracle_estimate_similarity_all_points.restype = C.c_int
        ok = L.rfd_oracle_estimate_similarity_all_points(s, d, s.shape[0], M)
        return M.reshape(2, 3) if ok else None
    inl = np.zeros(s.shape[0], np.uint8)
    L.rfd_oracle_estimate_similarity_lmeds.argtypes = list(L.rfd_oracle_estimate_similarity.argtypes) + [np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")]
    L.rfd_oracle_estimate_similarity_lmeds.restype = C.c_int
    ok = L.rfd_oracle_estimate_similarity_lmeds(s, d, s.shape[0], M, inl)
    if return_inliers:
        return (M.reshape(2, 3) if ok else None), inl.astype(bool)
    return M.reshape(2, 3) if ok else None


def lmeds_samples(n):
    """the two-point index pairs OpenCV's LMeDS draws for n points (cv::RNG re-seeded with (uint64)-1 on every call)"""
    L = lib()
    L.rfd_oracle_lmeds_samples.argtypes = [C.c_int, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"), C.c_int]
    L.rfd_oracle_lmeds_samples.restype = C.c_int
    p = np.zeros(128, np.int32)
    k = L.rfd_oracle_lmeds_samples(n, p, 64)
    return p[:2 * k].reshape(-1, 2)


def cv_ransac_num_iters(p, ep, model_points, max_iters):
    L = lib()
    L.rfd_oracle_cv_ransac_num_iters.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    L.rfd_oracle_cv_ransac_num_iters.restype = C.c_int
    return L.rfd_oracle_cv_ransac_num_iters(p, ep, model_points, max_iters)


def warp_affine(src, M, out_h, out_w):
    """cv::warpAffine(src, M, (out_w, out_h), INTER_LINEAR, BORDER_CONSTANT, 0) restated."""
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((out_h, out_w, 3), np.uint8)
    lib().rfd_oracle_warp_affine_u8c3(src, src.shape[0], src.shape[1], src.strides[0],
                                      np.ascontiguousarray(M, np.float64).reshape(6), dst, out_h, out_w)
    return dst


def face_alignment(src, box, kps, image_size=(112, 112), standard_landmarks=STANDARD_LANDMARKS):
    """FaceAlignment::call (face_alignment.rs:27-141) -> (crop [h,w,3] u8, status 0 warp / 1 fallback / -1 ROI error)."""
    src = np.ascontiguousarray(src, np.uint8)
    b = None if box is None else np.ascontiguousarray(box, np.float32)
    k = None if kps is None else np.ascontiguousarray(kps, np.float32).reshape(10)
    dst = np.zeros((image_size[1], image_size[0], 3), np.uint8)
    st = lib().rfd_oracle_face_alignment(src, src.shape[0], src.shape[1], src.strides[0],
                                         None if b is None else b.ctypes.data, None if k is None else k.ctypes.data,
                                         np.ascontiguousarray(standard_landmarks, np.float32).reshape(10),
                                         image_size[0], image_size[1], dst)
    return dst, st

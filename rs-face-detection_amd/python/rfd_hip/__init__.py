"""rfd_hip -- ctypes binding of librfd_hip.so (include/rfd.h) and a thin host mirror of the
reference's `RetinaFaceDetection` (src/pipeline/module/face_detection.rs:19-513).

All compute happens in the HIP library; this module only marshals numpy buffers.  There is no CPU
fallback: if the shared library is missing, or no MI355X is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "librfd_hip.so"))

RFD_OK = 0
RFD_ERR_INVALID_ARG = -1
RFD_ERR_NO_DEVICE = -2
PRECISION_BF16, PRECISION_F32 = 0, 1
RFD_ERR_HIP = -3
RFD_ERR_CAPACITY = -4
RFD_ERR_STATE = -5
RFD_ERR_IO = -6
RFD_ERR_COMM = -7
COMM_ID_BYTES = 128

BACKBONE_R50 = 0
BACKBONE_MNET025 = 1

STRIDES = (32, 16, 8)       # _feat_stride_fpn, face_detection.rs:52
NUM_ANCHORS = 2             # anchors per position


class RfdError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rfd status %d: %s" % (status, message))
        self.status = status


class rfd_config(C.Structure):
    _fields_ = [("image_w", C.c_int), ("image_h", C.c_int), ("max_batch_size", C.c_int),
                ("confidence_threshold", C.c_float), ("iou_threshold", C.c_float),
                ("device_id", C.c_int), ("max_det", C.c_int), ("max_src_w", C.c_int),
                ("max_src_h", C.c_int), ("backbone", C.c_int), ("precision", C.c_int), ("reserved", C.c_int * 5)]


class rfd_image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("height", C.c_int), ("width", C.c_int),
                ("stride", C.c_ssize_t)]


class rfd_dets(C.Structure):
    _fields_ = [("boxes", C.c_void_p), ("landmarks", C.c_void_p), ("count", C.c_void_p),
                ("total", C.c_void_p)]


class rfd_stats(C.Structure):
    _fields_ = [("ms_h2d", C.c_float), ("ms_preprocess", C.c_float), ("ms_network", C.c_float),
                ("ms_decode", C.c_float), ("ms_sort", C.c_float), ("ms_nms", C.c_float),
                ("ms_d2h", C.c_float), ("ms_total", C.c_float), ("candidates", C.c_int64),
                ("detections", C.c_int64), ("reserved", C.c_int64 * 4)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class rfd_layer_desc(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("cin", C.c_int), ("cout", C.c_int), ("kh", C.c_int),
                ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("has_affine", C.c_int),
                ("kind", C.c_int), ("reserved", C.c_int * 3)]


class rfd_op_desc(C.Structure):
    _fields_ = [("kind", C.c_int), ("layer", C.c_int), ("in_", C.c_int), ("out", C.c_int),
                ("out2", C.c_int), ("outf", C.c_int), ("res", C.c_int), ("relu", C.c_int),
                ("res_up2", C.c_int), ("res_post", C.c_int), ("head_softmax", C.c_int),
                ("y_coff", C.c_int), ("macs", C.c_double), ("in2", C.c_int), ("layer2", C.c_int),
                ("in_affine", C.c_int), ("layer_n2", C.c_int), ("x_coff", C.c_int), ("y_split", C.c_int),
                ("y_split_add", C.c_int), ("n_valid", C.c_int), ("branch", C.c_int), ("layer_b", C.c_int),
                ("out_b", C.c_int)]


class rfd_alignment_config(C.Structure):
    _fields_ = [("out_w", C.c_int32), ("out_h", C.c_int32), ("standard_landmarks", C.c_float * 10),
                ("reserved", C.c_int32 * 4)]


class rfd_tensor_desc(C.Structure):
    _fields_ = [("channels", C.c_int), ("height", C.c_int), ("width", C.c_int),
                ("is_f32", C.c_int), ("buffer", C.c_int), ("is_input", C.c_int),
                ("head_level", C.c_int), ("channels_logical", C.c_int), ("reserved", C.c_int * 3)]


# every symbol include/rfd.h declares (tests check that the library exports them all)
API_SYMBOLS = [
    "rfd_config_default", "rfd_create", "rfd_destroy", "rfd_last_error", "rfd_version",
    "rfd_graph_create", "rfd_graph_destroy", "rfd_graph_counts", "rfd_graph_layer", "rfd_graph_op",
    "rfd_graph_tensor", "rfd_graph_macs", "rfd_graph_workspace_bytes",
    "rfd_init_synthetic_weights", "rfd_num_layers", "rfd_get_layer_weights",
    "rfd_set_layer_weights", "rfd_get_layer_affine", "rfd_set_layer_affine", "rfd_detect_batch",
    "rfd_detect_batch_device", "rfd_sync", "rfd_set_stream", "rfd_preprocess", "rfd_forward", "rfd_decode_nms",
    "rfd_nms_sorted", "_nms", "rfd_get_stats", "rfd_get_config", "rfd_set_thresholds", "rfd_set_profiling",
    "rfd_get_conv_profile", "rfd_get_op_profile", "rfd_debug_tensor_io", "rfd_debug_run_ops", "rfd_debug_set_conv_tile", "rfd_debug_op_kernels", "rfd_debug_set_concurrency", "rfd_debug_persistent_kernel", "rfd_debug_poke_nms_flag", "rfd_selection_config_default",
    "rfd_select_faces", "rfd_detect_select_batch", "rfd_save_weights", "rfd_load_weights",
    "rfd_alignment_config_default", "rfd_align_faces", "rfd_detect_select_align_batch",
    "rfd_host_alloc", "rfd_host_free", "rfd_submit_batch", "rfd_collect_batch",
    "rfd_comm_get_unique_id", "rfd_comm_init", "rfd_comm_info", "rfd_gather_detections", "rfd_comm_destroy",
]

_lib = None


def load_library(path=None):
    """Load librfd_hip.so.  Raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RFD_HIP_LIB") or LIB_PATH  # RFD_HIP_LIB: A/B runs of two builds on one box
    if not os.path.exists(p):
        raise ImportError("librfd_hip.so not found at %s -- build it with "
                          "rs-face-detection_amd/build.sh (there is no CPU fallback)" % p)
    L = C.CDLL(p)
    vp, ci = C.c_void_p, C.c_int
    L.rfd_last_error.restype = C.c_char_p
    L.rfd_config_default.argtypes = [C.POINTER(rfd_config)]
    L.rfd_config_default.restype = None
    L.rfd_create.argtypes = [C.POINTER(rfd_config), C.POINTER(vp)]
    L.rfd_destroy.argtypes = [vp]
    L.rfd_destroy.restype = None
    L.rfd_graph_create.argtypes = [ci, ci, ci, C.POINTER(vp)]
    L.rfd_graph_destroy.argtypes = [vp]
    L.rfd_graph_destroy.restype = None
    L.rfd_graph_counts.argtypes = [vp] + [C.POINTER(ci)] * 4
    L.rfd_graph_layer.argtypes = [vp, ci, C.POINTER(rfd_layer_desc)]
    L.rfd_graph_op.argtypes = [vp, ci, C.POINTER(rfd_op_desc)]
    L.rfd_graph_tensor.argtypes = [vp, ci, C.POINTER(rfd_tensor_desc)]
    L.rfd_graph_macs.argtypes = [vp]
    L.rfd_graph_macs.restype = C.c_double
    L.rfd_graph_workspace_bytes.argtypes = [vp]
    L.rfd_graph_workspace_bytes.restype = C.c_double
    L.rfd_init_synthetic_weights.argtypes = [vp, C.c_uint64]
    L.rfd_num_layers.argtypes = [vp]
    L.rfd_get_layer_weights.argtypes = [vp, ci, vp, vp]
    L.rfd_set_layer_weights.argtypes = [vp, ci, vp, vp]
    L.rfd_get_layer_affine.argtypes = [vp, ci, vp, vp]
    L.rfd_set_layer_affine.argtypes = [vp, ci, vp, vp]
    L.rfd_detect_batch.argtypes = [vp, C.POINTER(rfd_image), ci, C.POINTER(rfd_dets)]
    L.rfd_detect_batch_device.argtypes = [vp, C.POINTER(rfd_image), ci, C.POINTER(rfd_dets), ci]
    L.rfd_sync.argtypes = [vp]
    L.rfd_set_stream.argtypes = [vp, vp]
    L.rfd_preprocess.argtypes = [vp, C.POINTER(rfd_image), ci, vp, vp, vp]
    L.rfd_forward.argtypes = [vp, vp, ci, C.POINTER(vp)]
    L.rfd_decode_nms.argtypes = [vp, C.POINTER(vp), ci, vp, C.POINTER(rfd_dets), vp]
    L.rfd_nms_sorted.argtypes = [vp, vp, C.POINTER(ci), vp, ci, ci, C.c_float]
    L._nms.argtypes = [vp, C.POINTER(ci), vp, ci, ci, C.c_float, ci]
    L._nms.restype = None
    L.rfd_get_stats.argtypes = [vp, C.POINTER(rfd_stats)]
    L.rfd_get_config.argtypes = [vp, C.POINTER(rfd_config)]
    L.rfd_set_thresholds.argtypes = [vp, C.c_float, C.c_float]
    L.rfd_set_profiling.argtypes = [vp, ci]
    L.rfd_get_conv_profile.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(ci)]
    L.rfd_get_op_profile.argtypes = [vp, vp, ci]
    L.rfd_debug_tensor_io.argtypes = [vp, ci, ci, vp, ci]
    L.rfd_debug_run_ops.argtypes = [vp, ci, ci, ci]
    L.rfd_debug_set_conv_tile.argtypes = [vp, ci]
    L.rfd_debug_set_concurrency.argtypes = [vp, ci, ci, ci, ci]
    L.rfd_debug_op_kernels.argtypes = [vp, ci, ci, ci, C.c_char_p, ci]
    L.rfd_debug_poke_nms_flag.argtypes = [vp, ci]
    L.rfd_debug_persistent_kernel.argtypes = [ci, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
    L.rfd_save_weights.argtypes = [vp, C.c_char_p]
    L.rfd_load_weights.argtypes = [vp, C.c_char_p]
    L.rfd_selection_config_default.argtypes = [vp]
    L.rfd_selection_config_default.restype = None
    L.rfd_select_faces.argtypes = [vp, C.POINTER(rfd_dets), vp, vp, ci, vp, ci, vp, vp, vp]
    L.rfd_detect_select_batch.argtypes = [vp, C.POINTER(rfd_image), ci, vp, ci, vp, vp, vp]
    L.rfd_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.rfd_host_free.argtypes = [vp]
    L.rfd_submit_batch.argtypes = [vp, C.POINTER(rfd_image), ci]
    L.rfd_collect_batch.argtypes = [vp, C.POINTER(rfd_dets), C.POINTER(ci)]
    L.rfd_comm_get_unique_id.argtypes = [vp]
    L.rfd_comm_init.argtypes = [vp, vp, ci, ci]
    L.rfd_comm_info.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    L.rfd_gather_detections.argtypes = [vp, C.POINTER(rfd_dets), ci, C.POINTER(rfd_dets)]
    L.rfd_comm_destroy.argtypes = [vp]
    L.rfd_alignment_config_default.argtypes = [C.POINTER(rfd_alignment_config)]
    L.rfd_alignment_config_default.restype = None
    L.rfd_align_faces.argtypes = [vp, C.POINTER(rfd_image), ci, vp, vp, vp, vp, vp, vp]
    L.rfd_detect_select_align_batch.argtypes = [vp, C.POINTER(rfd_image), ci, vp, ci, vp, vp, vp, vp, vp, vp]
    if path is None:
        _lib = L
    return L


def _check(status):
    if status < 0:
        raise RfdError(status, load_library().rfd_last_error().decode("utf-8", "replace"))
    return status


def head_shapes(n, net_h, net_w):
    """Shapes of the 9 head tensors (the reference's Triton output contract,
    face_detection.rs:286-312): per level 32,16,8: cls [n,2A,h,w], bbox [n,4A,h,w], lmk [n,10A,h,w]."""
    out = []
    for s in STRIDES:
        h, w = net_h // s, net_w // s
        out += [(n, 2 * NUM_ANCHORS, h, w), (n, 4 * NUM_ANCHORS, h, w), (n, 10 * NUM_ANCHORS, h, w)]
    return out


class Graph:
    """Host-only description of the build-defined network graph (no GPU needed)."""

    def __init__(self, backbone=BACKBONE_R50, image_w=640, image_h=640):
        self._L = load_library()
        self._g = C.c_void_p()
        _check(self._L.rfd_graph_create(backbone, image_w, image_h, C.byref(self._g)))
        n = [C.c_int() for _ in range(4)]
        _check(self._L.rfd_graph_counts(self._g, *[C.byref(x) for x in n]))
        self.num_layers, self.num_ops, self.num_tensors, self.num_buffers = [x.value for x in n]
        self.layers, self.ops, self.tensors = [], [], []
        for i in range(self.num_layers):
            d = rfd_layer_desc()
            _check(self._L.rfd_graph_layer(self._g, i, C.byref(d)))
            self.layers.append(d)
        for i in range(self.num_ops):
            d = rfd_op_desc()
            _check(self._L.rfd_graph_op(self._g, i, C.byref(d)))
            self.ops.append(d)
        for i in range(self.num_tensors):
            d = rfd_tensor_desc()
            _check(self._L.rfd_graph_tensor(self._g, i, C.byref(d)))
            self.tensors.append(d)
        self.macs = self._L.rfd_graph_macs(self._g)
        self.workspace_bytes = self._L.rfd_graph_workspace_bytes(self._g)

    def __del__(self):
        if getattr(self, "_g", None):
            self._L.rfd_graph_destroy(self._g)
            self._g = None


class FaceDetectionConfig:
    """Mirror of FaceDetectionConfig::new (src/pipeline/face_pipeline/config.rs:23-32)."""

    def __init__(self):
        self.model_name = "face_detection_retina"
        self.image_size = (640, 640)     # (w, h)
        self.max_batch_size = 1
        self.confidence_threshold = 0.7
        self.iou_threshold = 0.45
        self.timeout = 20


class RetinaFaceDetection:
    """Host mirror of the reference's RetinaFaceDetection (face_detection.rs:19-513).

    new(...)  -> __init__   (the Triton client / model config / model name arguments are gone)
    call      -> call       (one HxWx3 u8 BGR frame -> (det [K,5], kps [K,5,2]))
    plus call_batch and the stage-level entry points used by the parity tests.
    """

    def __init__(self, image_size=(640, 640), max_batch_size=1, confidence_threshold=0.7,
                 iou_threshold=0.45, device_id=0, max_det=1024, backbone=BACKBONE_R50, precision=0):
        self._L = load_library()
        cfg = rfd_config()
        self._L.rfd_config_default(C.byref(cfg))
        cfg.image_w, cfg.image_h = int(image_size[0]), int(image_size[1])
        cfg.max_batch_size = int(max_batch_size)
        cfg.confidence_threshold = float(confidence_threshold)
        cfg.iou_threshold = float(iou_threshold)
        cfg.precision = int(precision)   # 0 = bf16 product path, 1 = PRECISION_F32 parity mode (R50 only)
        cfg.device_id = int(device_id)
        cfg.max_det = int(max_det)
        cfg.backbone = int(backbone)
        self.cfg = cfg
        self._ctx = C.c_void_p()
        _check(self._L.rfd_create(C.byref(cfg), C.byref(self._ctx)))
        self.image_size = (cfg.image_w, cfg.image_h)
        self.max_det = cfg.max_det

    def close(self):
        for p in getattr(self, "_pinned", []):
            self._L.rfd_host_free(p)
        self._pinned = []
        if getattr(self, "_ctx", None):
            self._L.rfd_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.close()

    # ---- weights ----
    def init_synthetic_weights(self, seed=1234):
        _check(self._L.rfd_init_synthetic_weights(self._ctx, seed))

    def save_weights(self, path):
        _check(self._L.rfd_save_weights(self._ctx, os.fsencode(path)))

    def load_weights(self, path):
        _check(self._L.rfd_load_weights(self._ctx, os.fsencode(path)))

    def get_layer(self, idx, desc):
        w = np.zeros((desc.cout, desc.kh, desc.kw, desc.cin), np.float32)  # depthwise: cin = 1
        b = np.zeros(desc.cout, np.float32)
        _check(self._L.rfd_get_layer_weights(self._ctx, idx, w.ctypes.data, b.ctypes.data))
        return w, b

    def set_layer(self, idx, w, b):
        w = np.ascontiguousarray(w, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        _check(self._L.rfd_set_layer_weights(self._ctx, idx, w.ctypes.data, b.ctypes.data))

    def get_affine(self, idx, cout):
        s = np.zeros(cout, np.float32)
        t = np.zeros(cout, np.float32)
        _check(self._L.rfd_get_layer_affine(self._ctx, idx, s.ctypes.data, t.ctypes.data))
        return s, t

    def set_affine(self, idx, scale, shift):
        s = np.ascontiguousarray(scale, np.float32)
        t = np.ascontiguousarray(shift, np.float32)
        _check(self._L.rfd_set_layer_affine(self._ctx, idx, s.ctypes.data, t.ctypes.data))

    # ---- helpers ----
    @staticmethod
    def _images(frames):
        arr = (rfd_image * len(frames))()
        keep = []
        for i, f in enumerate(frames):
            if not (isinstance(f, np.ndarray) and f.dtype == np.uint8 and f.ndim == 3 and f.shape[2] == 3):
                raise RfdError(RFD_ERR_INVALID_ARG, "frames must be HxWx3 uint8 arrays (BGR)")
            if f.strides[2] != 1 or f.strides[1] != 3:
                f = np.ascontiguousarray(f)
            keep.append(f)
            arr[i].data = f.ctypes.data
            arr[i].height, arr[i].width = f.shape[0], f.shape[1]
            arr[i].stride = f.strides[0]
        return arr, keep

    def _alloc_dets(self, n):
        md = self.max_det
        boxes = np.zeros((n, md, 5), np.float32)
        lmk = np.zeros((n, md, 5, 2), np.float32)
        count = np.zeros(n, np.int32)
        total = np.zeros(n, np.int32)
        d = rfd_dets(boxes.ctypes.data, lmk.ctypes.data, count.ctypes.data, total.ctypes.data)
        return d, boxes, lmk, count, total

    @staticmethod
    def _split(boxes, lmk, count):
        return [(boxes[i, :count[i]].copy(), lmk[i, :count[i]].copy()) for i in range(len(count))]

    # ---- the hot path ----
    def call_batch(self, frames):
        """Batch form of `call`: list of HxWx3 u8 BGR frames -> list of (det [K,5], kps [K,5,2])."""
        arr, keep = self._images(frames)
        d, boxes, lmk, count, total = self._alloc_dets(len(frames))
        _check(self._L.rfd_detect_batch(self._ctx, arr, len(frames), C.byref(d)))
        self.last_total = total
        return self._split(boxes, lmk, count)

    # ---- pipelined host entry: the PCIe copy of batch i+1 overlaps the compute of batch i ----
    def host_frames(self, n, h, w):
        """n frames of h x w x 3 u8 in page-locked memory (rfd_host_alloc): decode into it, then submit()."""
        p = C.c_void_p()
        _check(self._L.rfd_host_alloc(n * h * w * 3, C.byref(p)))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_uint8 * (n * h * w * 3)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(n, h, w, 3)

    def submit(self, frames):
        arr, keep = self._images(frames)
        self._inflight = getattr(self, "_inflight", [])
        _check(self._L.rfd_submit_batch(self._ctx, arr, len(frames)))
        self._inflight.append((len(frames), keep))  # the frames must outlive the batch

    def collect(self):
        n, _ = self._inflight[0] if getattr(self, "_inflight", None) else (self.cfg.max_batch_size, None)
        d, boxes, lmk, count, total = self._alloc_dets(n)
        got = C.c_int(0)
        _check(self._L.rfd_collect_batch(self._ctx, C.byref(d), C.byref(got)))
        self._inflight.pop(0)
        self.last_total = total
        return self._split(boxes[:got.value], lmk[:got.value], count[:got.value])

    def call(self, image, is_debug=None):
        """RetinaFaceDetection::call (face_detection.rs:496): -> (det [K,5], kps [K,5,2])."""
        return self.call_batch([image])[0]

    def detect_device(self, frame_ptrs, shapes, out_boxes_ptr, out_lmk_ptr, out_count_ptr,
                      out_total_ptr, async_=False):
        """Frames and outputs already in device memory (raw device pointers)."""
        n = len(frame_ptrs)
        arr = (rfd_image * n)()
        for i, (p, (h, w)) in enumerate(zip(frame_ptrs, shapes)):
            arr[i].data, arr[i].height, arr[i].width, arr[i].stride = p, h, w, w * 3
        d = rfd_dets(out_boxes_ptr, out_lmk_ptr, out_count_ptr, out_total_ptr)
        _check(self._L.rfd_detect_batch_device(self._ctx, arr, n, C.byref(d), int(async_)))  # 0 sync, 1 async, 2 overlapped

    def sync(self):
        _check(self._L.rfd_sync(self._ctx))

    # ---- multi-GPU: RCCL all-gather of the detection slabs behind the C ABI ----
    @staticmethod
    def comm_unique_id():
        """rank 0: 128 opaque bytes to hand to every rank (any transport)."""
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        _check(load_library().rfd_comm_get_unique_id(C.addressof(buf)))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(self._L.rfd_comm_init(self._ctx, C.addressof(buf), int(rank), int(world)))

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        _check(self._L.rfd_comm_info(self._ctx, C.byref(r), C.byref(w)))
        return r.value, w.value

    def gather_detections(self, local_ptrs, n_local, all_ptrs):
        """local_ptrs / all_ptrs: (boxes, landmarks, count, total) DEVICE addresses; enqueued on the context's stream."""
        a = rfd_dets(*local_ptrs)
        b = rfd_dets(*all_ptrs)
        _check(self._L.rfd_gather_detections(self._ctx, C.byref(a), int(n_local), C.byref(b)))

    def comm_destroy(self):
        _check(self._L.rfd_comm_destroy(self._ctx))

    def set_stream(self, hip_stream):
        """Run on a caller-owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream); None restores."""
        _check(self._L.rfd_set_stream(self._ctx, hip_stream))

    # ---- next stage: FaceSelection::call (face_selection.rs:72-189) ----
    @staticmethod
    def _sel_cfg(cfg):
        c = (C.c_float * 4)(0.3, 0.3, 0.1, 0.0075)  # FaceSelectionConfig::new, config.rs:107-117
        if cfg is not None:
            for i, v in enumerate(cfg):
                c[i] = v
        return c

    @staticmethod
    def _sel_out(box, kps, found):
        return [(box[i].copy() if found[i] & 1 else None, kps[i].reshape(5, 2).copy() if found[i] == 3 else None)
                for i in range(len(found))]

    def select_faces(self, dets, sizes, is_enroll=False, cfg=None):
        """dets: list of (det [K,5], kps [K,5,2]) per image; sizes: list of (h, w) of the source frames."""
        n = len(dets)
        d, boxes, lmk, count, total = self._alloc_dets(n)
        for i, (a, b) in enumerate(dets):
            k = min(len(a), self.max_det)
            boxes[i, :k] = a[:k]
            lmk[i, :k] = b[:k]
            count[i] = k
        hh = np.array([s[0] for s in sizes], np.int32)
        ww = np.array([s[1] for s in sizes], np.int32)
        ob, ok, fd = np.zeros((n, 5), np.float32), np.zeros((n, 10), np.float32), np.zeros(n, np.int32)
        c = self._sel_cfg(cfg)
        _check(self._L.rfd_select_faces(self._ctx, C.byref(d), hh.ctypes.data, ww.ctypes.data, n, C.addressof(c),
                                        1 if is_enroll else 0, ob.ctypes.data, ok.ctypes.data, fd.ctypes.data))
        return self._sel_out(ob, ok, fd)

    def detect_select(self, frames, is_enroll=False, cfg=None):
        """FacePipeline::extract lines 198-208: detect, then return only the selected face of each frame."""
        arr, keep = self._images(frames)
        n = len(frames)
        ob, ok, fd = np.zeros((n, 5), np.float32), np.zeros((n, 10), np.float32), np.zeros(n, np.int32)
        c = self._sel_cfg(cfg)
        _check(self._L.rfd_detect_select_batch(self._ctx, arr, n, C.addressof(c), 1 if is_enroll else 0,
                                               ob.ctypes.data, ok.ctypes.data, fd.ctypes.data))
        return self._sel_out(ob, ok, fd)

    # ---- the stage after selection: FaceAlignment::call (face_alignment.rs:27-141) ----
    def _align_cfg(self, image_size=None, standard_landmarks=None):
        c = rfd_alignment_config()
        self._L.rfd_alignment_config_default(C.byref(c))  # FaceAlignmentConfig::new, config.rs:44-56
        if image_size is not None:
            c.out_w, c.out_h = int(image_size[0]), int(image_size[1])
        if standard_landmarks is not None:
            for i, v in enumerate(np.asarray(standard_landmarks, np.float32).reshape(10)):
                c.standard_landmarks[i] = v
        return c

    def align_faces(self, frames, selected, image_size=None, standard_landmarks=None):
        """frames: list of HxWx3 u8 BGR; selected: list of (box [5] or None, kps [5,2] or None) as select_faces /
        detect_select return them.  -> crops [n, out_h, out_w, 3] u8, status [n] (see rfd.h)."""
        arr, keep = self._images(frames)
        n = len(frames)
        ob, ok, fd = np.zeros((n, 5), np.float32), np.zeros((n, 10), np.float32), np.zeros(n, np.int32)
        for i, (b, k) in enumerate(selected):
            if b is not None:
                ob[i] = b
                fd[i] |= 1
            if k is not None:
                ok[i] = np.asarray(k, np.float32).reshape(10)
                fd[i] |= 2
        c = self._align_cfg(image_size, standard_landmarks)
        crops = np.zeros((n, c.out_h, c.out_w, 3), np.uint8)
        status = np.zeros(n, np.int32)
        _check(self._L.rfd_align_faces(self._ctx, arr, n, ob.ctypes.data, ok.ctypes.data, fd.ctypes.data, C.addressof(c),
                                       crops.ctypes.data, status.ctypes.data))
        return crops, status

    def detect_select_align(self, frames, is_enroll=False, sel_cfg=None, image_size=None, standard_landmarks=None):
        """FacePipeline::extract lines 198-216: detect, select, align -> (selected list, crops, status)."""
        arr, keep = self._images(frames)
        n = len(frames)
        ob, ok, fd = np.zeros((n, 5), np.float32), np.zeros((n, 10), np.float32), np.zeros(n, np.int32)
        sc = self._sel_cfg(sel_cfg)
        c = self._align_cfg(image_size, standard_landmarks)
        crops = np.zeros((n, c.out_h, c.out_w, 3), np.uint8)
        status = np.zeros(n, np.int32)
        _check(self._L.rfd_detect_select_align_batch(self._ctx, arr, n, C.addressof(sc), 1 if is_enroll else 0,
                                                     C.addressof(c), ob.ctypes.data, ok.ctypes.data, fd.ctypes.data,
                                                     crops.ctypes.data, status.ctypes.data))
        return self._sel_out(ob, ok, fd), crops, status

    # ---- stage-level entry points ----
    def preprocess(self, frames):
        """_preprocess + tensorise: -> det_img [n,H,W,3] u8, tensor [n,3,H,W] f32, det_scale [n]."""
        arr, keep = self._images(frames)
        n = len(frames)
        w, h = self.image_size
        det_img = np.zeros((n, h, w, 3), np.uint8)
        tensor = np.zeros((n, 3, h, w), np.float32)
        scale = np.zeros(n, np.float32)
        _check(self._L.rfd_preprocess(self._ctx, arr, n, det_img.ctypes.data, tensor.ctypes.data,
                                      scale.ctypes.data))
        return det_img, tensor, scale

    def forward(self, tensor):
        """tensor [n,3,H,W] f32 -> the 9 head tensors (f32, NCHW)."""
        t = np.ascontiguousarray(tensor, np.float32)
        n = t.shape[0]
        w, h = self.image_size
        assert t.shape == (n, 3, h, w)
        heads = [np.zeros(s, np.float32) for s in head_shapes(n, h, w)]
        ptrs = (C.c_void_p * 9)(*[x.ctypes.data for x in heads])
        _check(self._L.rfd_forward(self._ctx, t.ctypes.data, n, ptrs))
        return heads

    def decode_nms(self, heads, det_scale, want_gidx=False):
        """9 head tensors [n,C,h,w] + det_scale [n] -> list of (det, kps[, gidx]) per image."""
        hs = [np.ascontiguousarray(x, np.float32) for x in heads]
        n = hs[0].shape[0]
        w, h = self.image_size
        for x, s in zip(hs, head_shapes(n, h, w)):
            if x.shape != s:
                raise RfdError(RFD_ERR_INVALID_ARG, "head shape %s != %s" % (x.shape, s))
        sc = np.ascontiguousarray(det_scale, np.float32).reshape(n)
        ptrs = (C.c_void_p * 9)(*[x.ctypes.data for x in hs])
        d, boxes, lmk, count, total = self._alloc_dets(n)
        gidx = np.zeros((n, self.max_det), np.int32)
        _check(self._L.rfd_decode_nms(self._ctx, ptrs, n, sc.ctypes.data, C.byref(d),
                                      gidx.ctypes.data if want_gidx else None))
        self.last_total = total
        res = self._split(boxes, lmk, count)
        if want_gidx:
            res = [(a, b, gidx[i, :count[i]].copy()) for i, (a, b) in enumerate(res)]
        return res

    def nms_sorted(self, boxes, thresh):
        """Greedy NMS on rows pre-sorted by score descending -> kept row indices."""
        b = np.ascontiguousarray(boxes, np.float32)
        keep = np.zeros(max(b.shape[0], 1), np.int32)
        num = C.c_int(0)
        _check(self._L.rfd_nms_sorted(self._ctx, keep.ctypes.data, C.byref(num), b.ctypes.data,
                                      b.shape[0], b.shape[1] if b.ndim == 2 else 5, float(thresh)))
        return keep[:num.value].copy()

    # ---- test hooks ----
    def debug_write(self, tensor_id, arr):
        a = np.ascontiguousarray(arr)
        _check(self._L.rfd_debug_tensor_io(self._ctx, tensor_id, a.shape[0], a.ctypes.data, 1))

    def debug_read(self, tensor_id, n, desc):
        dt = np.float32 if desc.is_f32 else np.uint16
        a = np.zeros((n, desc.height, desc.width, desc.channels), dt)
        _check(self._L.rfd_debug_tensor_io(self._ctx, tensor_id, n, a.ctypes.data, 0))
        return a

    def debug_set_conv_tile(self, tile):
        _check(self._L.rfd_debug_set_conv_tile(self._ctx, tile))

    def debug_op_kernels(self, n, op, co_running=True):
        """kernel(s) the library would run op `op` with at n images per chain (list of names, no rfd:: prefix); nothing is launched"""
        buf = C.create_string_buffer(512)
        _check(self._L.rfd_debug_op_kernels(self._ctx, n, op, 1 if co_running else 0, buf, 512))
        return buf.value.decode().split(" + ")

    def debug_set_concurrency(self, multi_stream=True, split_min_part=8, split_max_parts=2, use_graph=True):
        _check(self._L.rfd_debug_set_concurrency(self._ctx, int(multi_stream), int(split_min_part), int(split_max_parts),
                                                 int(use_graph)))

    def debug_run(self, n, first_op, last_op):
        _check(self._L.rfd_debug_run_ops(self._ctx, n, first_op, last_op))

    # ---- introspection ----
    def stats(self):
        s = rfd_stats()
        _check(self._L.rfd_get_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    def set_thresholds(self, confidence_threshold, iou_threshold):
        _check(self._L.rfd_set_thresholds(self._ctx, confidence_threshold, iou_threshold))

    def set_profiling(self, enable):
        _check(self._L.rfd_set_profiling(self._ctx, 1 if enable else 0))

    def conv_profile(self):
        ms, fl, n = C.c_float(), C.c_double(), C.c_int()
        _check(self._L.rfd_get_conv_profile(self._ctx, C.byref(ms), C.byref(fl), C.byref(n)))
        return ms.value, fl.value, n.value

    def op_profile(self, nops):
        ms = np.zeros(nops, np.float32)
        _check(self._L.rfd_get_op_profile(self._ctx, ms.ctypes.data, nops))
        return ms

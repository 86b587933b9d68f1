/*
 * rfd.h -- C ABI of librfd_hip.so: the MI355X-native (gfx950) replacement of the detection stage of
 * okieraised/rs-face-detection, i.e. of
 *     RetinaFaceDetection::call(&self, image:&Mat, is_debug) -> (Array2<f32>[K,5], Array3<f32>[K,5,2])
 *     (reference src/pipeline/module/face_detection.rs:496), invoked from
 *     FacePipeline::extract (src/pipeline/face_pipeline/pipeline.rs:198).
 *
 * The reference ships a 640x640 f32 tensor to a Triton server (face_detection.rs:279) and decodes
 * the 9 returned head tensors on the CPU.  Here preprocess, the RetinaFace network, decode, sort,
 * NMS and rescale all run on the GPU as HIP kernels; this header is what a Rust `extern "C"` block
 *  binds.  The only native ABI the reference itself declares is `_nms`
 * (src/rcnn/gpu_nms.hpp:7); a compatible symbol is exported too.
 *
 * Conventions (mirroring `_nms`): the caller allocates every input and output buffer; the context
 * owns device weights, workspaces, streams.  Every function returns 0 on success or a negative
 * rfd_status; rfd_last_error() returns a message for the last failure on the calling thread.
 * There is NO CPU fallback: without a usable HIP device rfd_create fails with RFD_ERR_NO_DEVICE.
 * A context is not re-entrant (one stream + workspace); use one context per thread.
 */
#ifndef RFD_H
#define RFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFD_VERSION 1

#if defined(__GNUC__)
#define RFD_API __attribute__((visibility("default")))
#else
#define RFD_API
#endif

typedef enum rfd_status {
    RFD_OK = 0,
    RFD_ERR_INVALID_ARG = -1, /* null pointer, bad shape, channels != 3 (reference: OpenCV error / at_2d failure, face_detection.rs:137,226) */
    RFD_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at creation */
    RFD_ERR_HIP = -3,         /* a HIP call or kernel failed, or a device-side bounded wait gave up (chunked NMS, ring convolution): the
                               * detections since the last synchronisation are invalid (reference: Triton RPC failure, face_detection.rs:282) */
    RFD_ERR_CAPACITY = -4,    /* batch / frame / detections exceed the configured capacity */
    RFD_ERR_STATE = -5,       /* weights not initialised (reference: empty model config, face_detection.rs:239) */
    RFD_ERR_IO = -6,          /* weight file could not be read / written */
    RFD_ERR_COMM = -7         /* librccl missing, or an RCCL call failed (multi-GPU gather) */
} rfd_status;

typedef enum rfd_backbone {
    RFD_BACKBONE_R50 = 0,     /* RetinaFace ResNet-50 + FPN + SSH */
    RFD_BACKBONE_MNET025 = 1  /* RetinaFace MobileNet-0.25 (same head/anchor contract) */
} rfd_backbone;

/*
 * Mirrors FaceDetectionConfig (src/pipeline/face_pipeline/config.rs:13-33) and the arguments of
 * RetinaFaceDetection::new (face_detection.rs:41-49): image_size is (w,h); the Triton client /
 * model-config / model-name arguments are gone (the network runs in-process); device, batch
 * capacity and output capacity are new.
 */
typedef enum rfd_precision { RFD_PRECISION_BF16 = 0, RFD_PRECISION_F32 = 1 } rfd_precision;

typedef struct rfd_config {
    int image_w;                /* config.rs:26 image_size.0, default 640 */
    int image_h;                /* config.rs:26 image_size.1, default 640 */
    int max_batch_size;         /* config.rs:28 (reference: 1); frames per rfd_detect_batch call */
    float confidence_threshold; /* config.rs:29, default 0.7; keep rows with score >= threshold */
    float iou_threshold;        /* config.rs:30, default 0.45; suppress on IoU > threshold */
    int device_id;              /* HIP device ordinal */
    int max_det;                /* capacity (rows per image) of the output slabs; default 1024 */
    int max_src_w;              /* largest source frame the context must stage; default 3840 */
    int max_src_h;              /* default 2160 */
    int backbone;               /* rfd_backbone */
    int precision;              /* rfd_precision: 0 = bf16 activations / weights with f32 accumulation (the product path,
                                 * BASELINE.json configs[2]); 1 = the f32 parity mode (RetinaFace-R50 only): f32 weights
                                 * and activations, convolutions accumulated in f64 and rounded once per output -- the reference's
                                 * FP32 tensor contract (face_detection.rs:261), reproducible to the bit by any evaluation that
                                 * sums the same products exactly.  A correctness mode: plain FMA kernels, one stream, ~60x slower
                                 * than the bf16 path. */
    int reserved[5];
} rfd_config;

/* A decoded source frame: HxWx3 u8, BGR, row stride in bytes (an OpenCV Mat CV_8UC3,
 * as produced by byte_data_to_opencv, src/utils/utils.rs:8-52). */
typedef struct rfd_image {
    const uint8_t *data;
    int height;
    int width;
    ptrdiff_t stride;
} rfd_image;

/*
 * Caller-allocated outputs for n frames.  Row layout equals the reference's return value
 * (face_detection.rs:432-464, 473-493): boxes row = x1,y1,x2,y2,score in SOURCE-image pixels,
 * rows in descending score (kept order); landmarks row = 5 x (x,y).
 *   boxes     [n][max_det][5]  f32
 *   landmarks [n][max_det][10] f32
 *   count     [n] i32 = min(K, max_det)
 *   total     [n] i32 = K, the untruncated number of detections (may be NULL)
 */
typedef struct rfd_dets {
    float *boxes;
    float *landmarks;
    int32_t *count;
    int32_t *total;
} rfd_dets;

/* Per-call stage timings measured with HIP events on the context's stream (milliseconds). */
typedef struct rfd_stats {
    float ms_h2d;
    float ms_preprocess;
    float ms_network;
    float ms_decode;
    float ms_sort;
    float ms_nms;
    float ms_d2h;
    float ms_total;
    int64_t candidates; /* rows with score >= threshold, summed over the batch */
    int64_t detections; /* kept rows, summed over the batch (host-output entry points only: the device-resident ones leave the counts in HBM and do not read them back) */
    int64_t reserved[4];
} rfd_stats;

typedef struct rfd_ctx rfd_ctx;

/* ---- lifecycle (replaces RetinaFaceDetection::new, face_detection.rs:41-129, and the Triton
 *      channel set-up of FacePipeline::new, pipeline.rs:64-128) ---- */
RFD_API void rfd_config_default(rfd_config *cfg);
RFD_API int rfd_create(const rfd_config *cfg, rfd_ctx **out);
RFD_API void rfd_destroy(rfd_ctx *ctx);
RFD_API const char *rfd_last_error(void);
RFD_API int rfd_version(void);

/* ---- network graph description (host only: works without a GPU).  The reference never sees the
 *      graph (it only knows the Triton model name, config.rs:25); these calls expose the
 *      build-defined graph so that tests can rebuild it layer by layer. ---- */
typedef struct rfd_graph rfd_graph;
typedef struct rfd_layer_desc {
    char name[64];
    int cin, cout, kh, kw, stride, pad;
    int has_affine; /* per-channel scale/shift + ReLU applied after the residual add (or the pool) */
    int kind;       /* 0 conv, 1 depthwise 3x3 (cin = 1, cout = channels; weights [C][3][3][1]), 2 first 3x3/2, 3 conv0 7x7/2 */
    int reserved[3];
} rfd_layer_desc;
typedef struct rfd_op_desc {
    int kind;  /* 0: conv0 7x7/2 + bias + ReLU, 1: maxpool 3x3/2 (+ affine + ReLU), 2: conv,
                  3: fused stem = kind 0 then kind 1 (conv0 result rounded to bf16 in between),
                  4: depthwise 3x3 + bias + ReLU, 5: first 3x3/2 conv (3 input channels) + bias + ReLU,
                  6: back-to-back pair: kind 2 (out = raw sum) followed by out_b = relu(conv(relu(affine(out)), layer_b)) */
    int layer; /* weights used (kind 1: the layer whose affine is applied) */
    int in, out, out2, outf, res; /* tensor ids, -1 = none: out = bf16 result, out2 = relu(affine(v)),
                                     outf = f32 result (heads), res = residual input */
    int relu, res_up2, res_post, head_softmax, y_coff;
    double macs; /* multiply-accumulates per image */
    int in2, layer2; /* fused 1x1 shortcut conv: + conv(tensor in2, layer2) (+ its bias); -1 = none */
    int in_affine;   /* >= 0: the input is first mapped through relu(x*scale+shift) of that layer's affine */
    int layer_n2;    /* >= 0: a sibling conv on the same input fused along N; its output channels follow */
    int x_coff;      /* the input is the channel slice [x_coff, x_coff + cin) of tensor `in` */
    int y_split, y_split_add; /* output channel n goes to y_coff + n (+ y_split_add if n >= y_split) */
    int n_valid;     /* only output channels < n_valid are stored */
    int branch;      /* 0 main chain; 1, 2: independent side chains that run on their own HIP streams */
    int layer_b, out_b; /* kind 6: the next unit's conv1 applied to relu(affine(out)), and its output tensor */
} rfd_op_desc;
typedef struct rfd_tensor_desc {
    int channels, height, width; /* channels = device channels: channels_logical zero-padded (to 64) */
    int is_f32, buffer, is_input, head_level; /* head_level: 1,2,3 = stride 32,16,8 head tensor */
    int channels_logical;
    int reserved[3];
} rfd_tensor_desc;
RFD_API int rfd_graph_create(int backbone, int image_w, int image_h, rfd_graph **out);
RFD_API void rfd_graph_destroy(rfd_graph *g);
RFD_API int rfd_graph_counts(const rfd_graph *g, int *layers, int *ops, int *tensors, int *buffers);
RFD_API int rfd_graph_layer(const rfd_graph *g, int idx, rfd_layer_desc *d);
RFD_API int rfd_graph_op(const rfd_graph *g, int idx, rfd_op_desc *d);
RFD_API int rfd_graph_tensor(const rfd_graph *g, int idx, rfd_tensor_desc *d);
RFD_API double rfd_graph_macs(const rfd_graph *g);            /* conv MACs per image */
RFD_API double rfd_graph_workspace_bytes(const rfd_graph *g); /* planned activation bytes per image */

/* ---- weights (replace Triton's model repository for "face_detection_retina", config.rs:25) ---- */
RFD_API int rfd_init_synthetic_weights(rfd_ctx *ctx, uint64_t seed);
RFD_API int rfd_num_layers(const rfd_ctx *ctx);
/* weights [cout][kh][kw][cin] f32 (values as stored on device, i.e. bf16-rounded), bias [cout];
 * BatchNorm is expected to be folded into weights/bias by the caller. */
RFD_API int rfd_get_layer_weights(rfd_ctx *ctx, int idx, float *weights, float *bias);
RFD_API int rfd_set_layer_weights(rfd_ctx *ctx, int idx, const float *weights, const float *bias);
/* Weight file (replaces Triton's model repository entry for the detector): little-endian; header "RFDW", u32
 * version = 1, u32 backbone, u32 n_layers; per layer: char name[64], i32 cin, cout, kh, kw, stride, pad, kind,
 * has_affine, then f32 weights [cout][kh][kw][cin], f32 bias [cout], and if has_affine f32 scale [cout], shift
 * [cout].  BatchNorm must be folded by the exporter.  rfd_load_weights checks every shape against the graph. */
RFD_API int rfd_save_weights(rfd_ctx *ctx, const char *path);
RFD_API int rfd_load_weights(rfd_ctx *ctx, const char *path);
/* the post-add affine (scale, shift per output channel) of layers with has_affine */
RFD_API int rfd_get_layer_affine(rfd_ctx *ctx, int idx, float *scale, float *shift);
RFD_API int rfd_set_layer_affine(rfd_ctx *ctx, int idx, const float *scale, const float *shift);

/* ---- the hot path: replaces RetinaFaceDetection::call (face_detection.rs:496-513) for a batch
 *      of frames.  Host buffers in, host buffers out; synchronous. ---- */
RFD_API int rfd_detect_batch(rfd_ctx *ctx, const rfd_image *imgs, int n, rfd_dets *out);

/* Same, with every `data` pointer of imgs[] and every pointer of `out` in DEVICE memory (frames
 * already resident in HBM, detections left in HBM for a following RCCL gather).  The imgs[] array
 * itself is host memory.  Enqueued on the context's stream; returns after the stream has drained
 * unless `async` is non-zero (then call rfd_sync before reading results / reusing buffers).
 * async = 1: the whole call is ordered on the context's stream (frames may be produced by earlier work on it).
 * async = 2: cross-call overlap -- the frames must be COMPLETE when the call is made (not still being written by
 *            enqueued work); the network chains then start without waiting for the previous call's decode / NMS, which
 *            stay on the context's stream together with this call's, so results, rfd_sync and a following collective
 *            on that stream behave as with async = 1.  Falls back to async = 1 for batches that are not split. */
RFD_API int rfd_detect_batch_device(rfd_ctx *ctx, const rfd_image *imgs, int n, rfd_dets *out, int async);
RFD_API int rfd_sync(rfd_ctx *ctx);
/* Enqueue on a caller-owned hipStream_t instead of the context's own stream (NULL restores it), so
 * that a following collective (RCCL gather of the detection slabs) is stream-ordered behind the
 * detector without a host synchronisation. */
RFD_API int rfd_set_stream(rfd_ctx *ctx, void *hip_stream);

/* ---- multi-GPU (the reference is single-device, one image per call: face_detection.rs:220).
 *      Frames shard image-parallel, one process (or thread) and one context per GPU, weights replicated; the only
 *      exchange step is the gather of the per-rank detection slabs, an RCCL all-gather over xGMI enqueued on the
 *      context's stream, i.e. stream-ordered behind the rank's NMS with no host synchronisation.
 *        rank 0:      rfd_comm_get_unique_id(id)          -> send the RFD_COMM_ID_BYTES bytes to every rank out of band
 *        every rank:  rfd_comm_init(ctx, id, rank, world) (collective: returns when all ranks have joined)
 *        per batch:   rfd_detect_batch_device(ctx, ..., &local, async) ; rfd_gather_detections(ctx, &local, n_local, &all)
 *      `local` and `all` are DEVICE slabs; every rank passes the same n_local (a short tail rank pads with count = 0) and
 *      `all` holds world * n_local frames, rank-major = the original frame order of a contiguous split:
 *      boxes [world*n_local][max_det][5], landmarks [..][max_det][10], count [..], total [..] (total may be NULL in both).
 *      librccl is loaded on first use (dlopen; RFD_RCCL_LIB overrides the name): single-GPU users never need it. ---- */
#define RFD_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
RFD_API int rfd_comm_get_unique_id(void *id /* RFD_COMM_ID_BYTES bytes, host */);
RFD_API int rfd_comm_init(rfd_ctx *ctx, const void *unique_id, int rank, int world);
RFD_API int rfd_comm_info(const rfd_ctx *ctx, int *rank, int *world); /* world = 0: no communicator */
RFD_API int rfd_gather_detections(rfd_ctx *ctx, const rfd_dets *local, int n_local, rfd_dets *all);
RFD_API int rfd_comm_destroy(rfd_ctx *ctx);

/* ---- stage-level entry points (parity tests; each mirrors one reference stage) ---- */

/* _preprocess + tensorise (face_detection.rs:131-198, 220-232) for n frames.
 * det_img [n][image_h][image_w][3] u8 (may be NULL), tensor [n][3][image_h][image_w] f32 R,G,B
 * planes raw 0..255 (may be NULL), det_scale [n] f32.  Host pointers. */
RFD_API int rfd_preprocess(rfd_ctx *ctx, const rfd_image *imgs, int n, uint8_t *det_img, float *tensor,
                   float *det_scale);

/* The network: tensor [n][3][image_h][image_w] f32 (the reference's Triton input contract,
 * face_detection.rs:220-277) -> 9 head tensors, f32, NCHW, in the reference's slot order
 * 32,16,8 x (cls [n,4,h,w], bbox [n,8,h,w], lmk [n,20,h,w]) (face_detection.rs:286-312, 319-407).
 * Host pointers. */
RFD_API int rfd_forward(rfd_ctx *ctx, const float *tensor, int n, float *const heads[9]);

/* Everything after the network (face_detection.rs:319-493) on caller-supplied head tensors
 * (same contract as rfd_forward's outputs), det_scale [n].  Host pointers.
 * gidx (may be NULL) [n][max_det] i32 receives the global anchor index of every kept row. */
RFD_API int rfd_decode_nms(rfd_ctx *ctx, const float *const heads[9], int n, const float *det_scale,
                   rfd_dets *out, int32_t *gidx);

/* Greedy NMS on boxes pre-sorted by score descending: the contract of the reference's (never
 * built) CUDA entry point `_nms` (src/rcnn/nms_kernel.cu:91-144, src/rcnn/gpu_nms.hpp:7) with the
 * live path's survivor rule (src/processing/nms.rs:58).  boxes: host [boxes_num][boxes_dim>=4]
 * f32; keep: capacity boxes_num.  Returns a status (the reference returns void and prints). */
RFD_API int rfd_nms_sorted(rfd_ctx *ctx, int32_t *keep, int *num_out, const float *boxes, int boxes_num,
                   int boxes_dim, float thresh);

/* Drop-in for the reference's declaration (gpu_nms.hpp:7): uses a process-wide context on
 * device_id, prints nothing, leaves *num_out = -1 on failure. */
RFD_API void _nms(int32_t *keep, int *num_out, float *boxes, int boxes_num, int boxes_dim, float thresh,
          int device_id);

/* ---- next stage: FaceSelection::call, face_selection.rs:72-189, as a device epilogue.
 *      Defaults = FaceSelectionConfig::new (config.rs:107-117).  found[i]: 0 = no face selected (reference:
 *      (None, None)), 1 = box only, 3 = box + key points.  Only the rows the detector returned (count <=
 *      max_det) take part. ---- */
typedef struct rfd_selection_config {
    float margin_center_left_ratio;  /* 0.3 */
    float margin_center_right_ratio; /* 0.3 */
    float margin_edge_ratio;         /* 0.1 */
    float minimum_face_ratio;        /* 0.0075 */
} rfd_selection_config;
RFD_API void rfd_selection_config_default(rfd_selection_config *cfg);
/* Stage-level: selection on host detections as returned by rfd_detect_batch (dets->boxes/landmarks/count),
 * img_h/img_w [n] = source frame sizes; out_box [n][5], out_kps [n][10], found [n].  Host pointers. */
RFD_API int rfd_select_faces(rfd_ctx *ctx, const rfd_dets *dets, const int *img_h, const int *img_w, int n,
                             const rfd_selection_config *cfg, int is_enroll, float *out_box, float *out_kps,
                             int32_t *found);
/* Fused: detect n frames (host buffers) and return only the selected face of each: detections stay in HBM,
 * 16 floats per frame cross PCIe.  = FacePipeline::extract lines 198-208 (pipeline.rs). */
RFD_API int rfd_detect_select_batch(rfd_ctx *ctx, const rfd_image *imgs, int n, const rfd_selection_config *cfg,
                                    int is_enroll, float *out_box, float *out_kps, int32_t *found);

/* ---- pipelined host entry (host decode + H2D staging, utils.rs:8-52 is the producer) ----
 *      rfd_detect_batch is synchronous, so the PCIe copy of a batch cannot overlap the compute of the previous one.
 *      rfd_submit_batch enqueues H2D (own copy stream) + the whole hot path + D2H of the detections and returns;
 *      rfd_collect_batch waits for the OLDEST submitted batch and fills `out` exactly as rfd_detect_batch would.
 *      Up to 2 batches may be in flight (a third submit returns RFD_ERR_STATE).  The frames must stay valid and
 *      unchanged until their batch has been collected.  For the copy to be a true asynchronous DMA the frames should
 *      live in page-locked memory: rfd_host_alloc / rfd_host_free hand it out (decode straight into it); pageable
 *      frames work but are copied synchronously by the runtime.  Not to be mixed with the other detect calls while a
 *      batch is in flight. ---- */
RFD_API int rfd_host_alloc(size_t bytes, void **ptr);
RFD_API int rfd_host_free(void *ptr);
RFD_API int rfd_submit_batch(rfd_ctx *ctx, const rfd_image *imgs, int n);
RFD_API int rfd_collect_batch(rfd_ctx *ctx, rfd_dets *out, int *n_out);

/* ---- FaceAlignment: FaceAlignment::call, src/pipeline/module/face_alignment.rs:27-141 --
 *      the step after selection in FacePipeline::extract (pipeline.rs:210-216).  For each frame the selected face is
 *      mapped onto the out_w x out_h template: 4-DOF similarity from its five key points to `standard_landmarks`
 *      as estimate_affine_partial_2d(LMEDS, 3.0, 2000, 0.99, 10) :48-60 computes it -- OpenCV's LMedS restated: 13
 *      two-point samples of the re-seeded cv::RNG, least median, inlier rule, least squares over the inliers (the
 *      fixed point of its Levenberg-Marquardt refinement; parity against a running OpenCV unpinned)
 *      + cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) :112-120 restated in integer
 *      arithmetic; degenerate key points take the reference's crop + resize branch :62-110.
 *      Defaults = FaceAlignmentConfig::new (config.rs:44-56): 112 x 112 and the ArcFace template.
 *      status[i]: 0 aligned, 1 crop + resize fallback, -1 no key points (the reference's call returns Err),
 *      -2 no face selected, -3 fallback ROI outside the frame (the reference's Mat::roi returns Err); crops with a
 *      negative status are zero-filled.  out_crops: [n][out_h][out_w][3] u8 BGR, host. ---- */
typedef struct rfd_alignment_config {
    int32_t out_w, out_h;            /* 112, 112 */
    float standard_landmarks[10];    /* x0,y0 ... x4,y4 */
    int32_t reserved[4];
} rfd_alignment_config;
RFD_API void rfd_alignment_config_default(rfd_alignment_config *cfg);
/* Stage-level: frames (host), selected boxes [n][5], key points [n][10] and flags found [n] as returned by
 * rfd_select_faces / rfd_detect_select_batch. */
RFD_API int rfd_align_faces(rfd_ctx *ctx, const rfd_image *imgs, int n, const float *boxes, const float *kps,
                            const int32_t *found, const rfd_alignment_config *cfg, uint8_t *out_crops,
                            int32_t *status);
/* Fused: detect + select + align n frames; the frames cross PCIe once, detections never leave HBM, 16 floats and one
 * crop per frame come back.  = FacePipeline::extract lines 198-216 (pipeline.rs). */
RFD_API int rfd_detect_select_align_batch(rfd_ctx *ctx, const rfd_image *imgs, int n,
                                          const rfd_selection_config *sel_cfg, int is_enroll,
                                          const rfd_alignment_config *align_cfg, float *out_box, float *out_kps,
                                          int32_t *found, uint8_t *out_crops, int32_t *status);

/* ---- introspection ---- */
RFD_API int rfd_get_stats(rfd_ctx *ctx, rfd_stats *stats);
RFD_API int rfd_get_config(const rfd_ctx *ctx, rfd_config *cfg);
/* confidence_threshold / iou_threshold are plain fields of the reference's struct
 * (face_detection.rs:26-27); they may be changed between calls. */
RFD_API int rfd_set_thresholds(rfd_ctx *ctx, float confidence_threshold, float iou_threshold);
/* Per-launch profiling of the network: when enabled, every network op is bracketed by HIP events on
 * the context's stream.  rfd_get_conv_profile returns, for the last synchronous detect/forward call,
 * the summed duration (ms) of the implicit-GEMM conv launches, the FLOPs they performed and their
 * number; rfd_get_op_profile copies the per-op durations (ms) and returns the op count. */
RFD_API int rfd_set_profiling(rfd_ctx *ctx, int enable);
RFD_API int rfd_get_conv_profile(rfd_ctx *ctx, float *ms_conv, double *flops_conv, int *launches);
RFD_API int rfd_get_op_profile(rfd_ctx *ctx, float *ms, int cap);

/* ---- test hooks (not part of the drop-in surface): raw access to a network tensor (device layout:
 *      NHWC, bf16 or f32 as rfd_tensor_desc says; n * C*H*W elements) and partial execution of the
 *      op list [first_op, last_op] (last_op < 0: to the end), so every op can be checked in isolation. */
RFD_API int rfd_debug_tensor_io(rfd_ctx *ctx, int tensor_id, int n, void *host, int write);
RFD_API int rfd_debug_run_ops(rfd_ctx *ctx, int n, int first_op, int last_op);
/* force the conv tile configuration: 0 = heuristic, 1 = 128-row four-wave tiles, 2 = 256x128 tiles where legal, 16 = the
 * weight-resident pair kernel for stage 1's pairs, 17 = the wave-specialised ring form wherever the layer shape allows, 18 = the
 * eight-wave 128x128 generic tile, 19 = the four-wave merged-kx 3x3 kernel (the full list: launch_conv in csrc/kernels_conv.hip) */
RFD_API int rfd_debug_set_conv_tile(rfd_ctx *ctx, int tile);
/* which kernel(s) op `op` of the network would be run by at `n` images per chain (co_running != 0: as one of the two chains of a
 * split pass) -- the names rocprofv3 reports without the rfd:: prefix, " + "-separated when an op takes two launches.  Nothing is
 * launched.  tools/traffic_model.py and tools/roof_gap.py attribute bytes and time to kernels through this call. */
RFD_API int rfd_debug_op_kernels(rfd_ctx *ctx, int n, int op, int co_running, char *names, int cap);
/* execution structure of the network pass: side streams for independent chains on/off; batch split into
 * clamp(n / split_min_part, 1, split_max_parts) contiguous parts that run as independent chains on their own streams
 * (split_max_parts <= 1 = never; at most 4); hipGraph replay of unsplit passes on/off.  Every structure gives
 * bit-identical results (tests/test_concurrency_gpu.py). */
RFD_API int rfd_debug_set_concurrency(rfd_ctx *ctx, int multi_stream, int split_min_part, int split_max_parts,
                                      int use_graph);

/* Sets the device word through which a chunk workgroup of the dense-crowd NMS -- or a wave of a ring convolution -- reports that
 * it gave up waiting (tests: the next call that synchronises must then fail with RFD_ERR_HIP and clear the word). */
RFD_API int rfd_debug_poke_nms_flag(rfd_ctx *ctx, int value);
/* The list of PERSISTENT kernels (one workgroup per CU, looping over work items) and the dynamic LDS every launch of one
 * requests -- always the CU's whole 160 KiB, so that no other kernel's workgroup can share the CU.
 * Needs no context and no GPU: tests/test_build_cpu.py walks it against the kernels of the code object.  Returns the number
 * of entries; fills name / lds_bytes for 0 <= i < that number. */
RFD_API int rfd_debug_persistent_kernel(int i, const char **name, size_t *lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* RFD_H */

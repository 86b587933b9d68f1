// kernels.h -- launch interfaces of the HIP kernels (all gfx950).
#pragma once
#include "rfd_common.h"

namespace rfd {

// ---------------------------------------------------------------- preprocess (kernels_pre.hip)
// One source frame and its letterbox geometry (reference face_detection.rs:131-198).
struct PreImage {
    const uint8_t *src; // device, HxWx3 u8 BGR
    long long stride;   // bytes per source row
    int h, w;
    int new_w, new_h;   // resized extent pasted at (0,0) of the canvas
    int area_fast;      // both scale factors exactly 2: OpenCV's INTER_LINEAR -> INTER_AREA switch
    int pad;
    double scale_x, scale_y; // 1 / (new_w / w), 1 / (new_h / h) in f64, as cv::resize computes them
};

struct PreParams {
    const PreImage *imgs; // device [n]
    int net_h, net_w;
    bf16_t *out_nhwc4;    // [n][net_h][net_w][4] bf16 R,G,B,0  (network input) or null
    uint8_t *out_det_img; // [n][net_h][net_w][3] u8 BGR canvas (reference det_img) or null
    float *out_tensor;    // [n][3][net_h][net_w] f32 R,G,B planes (reference Triton input) or null
};
int launch_preprocess(const PreParams &p, int n, hipStream_t s);
// [n][3][H][W] f32 planes -> [n][H][W][4] bf16 (used by rfd_forward, whose input is the tensor)
int launch_tensor_to_nhwc4(const float *tensor, bf16_t *out, int n, int H, int W, hipStream_t s);

// ---------------------------------------------------------------- decode / sort / NMS (kernels_post.hip)
struct DecodeParams {
    const float *cls[kNumLevels];
    const float *bbox[kNumLevels];
    const float *lmk[kNumLevels];
    int fh[kNumLevels], fw[kNumLevels], stride[kNumLevels], level_off[kNumLevels];
    float base_anchor[kNumLevels][kA][4];
    int total_anchors;
    int net_h, net_w;
    float conf_thr;
    float *rows;    // [n][total_anchors][16]
    uint64_t *keys; // [n][total_anchors]
    int *count;     // [n], zeroed before launch
};
int launch_decode(const DecodeParams &p, int n, bool nchw, hipStream_t s);

int launch_sort(uint64_t *keys, const int *count, const float *rows, uint64_t *sorted_keys,
                float4 *sorted_boxes, int total_anchors, int n, hipStream_t s);

struct NmsParams {
    const uint64_t *sorted_keys; // [n][total_anchors] (null with presorted boxes)
    const float4 *sorted_boxes;  // [n][total_anchors]
    const float *rows;           // [n][total_anchors][16] (null: emit indices only)
    const int *count;            // [n]
    const float *det_scale;      // [n]
    int presorted_n;             // >= 0: every image has exactly this many boxes (rfd_nms_sorted)
    int total_anchors;
    int max_det;
    int nwords_cap;              // filled by launch_nms
    float iou_thr;
    float *out_boxes;            // [n][max_det][5]
    float *out_lmk;              // [n][max_det][10]
    int *out_count;              // [n]
    int *out_total;              // [n]
    int *out_gidx;               // [n][max_det] or null
    // chunked kernel (dense crowds: kNmsChunks workgroups per image; all null / 0: one workgroup per image)
    float4 *kept_boxes;          // [n][total_anchors] scratch: the kept boxes of an image in score order, chunk after chunk
    int *chunk_state;            // [n][kNmsChunks][2]: {kept count, epoch of the launch that published it}; zero-initialised once
    int *spin_fail;              // set to 1 if a chunk gave up waiting for its predecessor (bounded spin; never expected)
    unsigned *ticket;            // two counters: workgroups draw their (image, chunk) from ticket[ticket_sel] in the order they start
    int ticket_sel;              // 0 / 1, alternating between the launches of a context (the idle counter is zeroed by ticket 0)
    int epoch;                   // > 0, different for every launch that uses chunk_state
};
constexpr int kNmsChunks = 4;
int launch_nms(NmsParams p, int n_images, hipStream_t s, bool *used_chunked = nullptr);

// FaceSelection::call on the device (face_selection.rs:72-189): per image, over its kept detections
struct SelectParams {
    const float *boxes; // [n][max_det][5]
    const float *lmk;   // [n][max_det][10]
    const int *count;   // [n]
    const int *img_h, *img_w; // [n] source frame sizes
    int n, max_det, is_enroll;
    float margin_center_left_ratio, margin_center_right_ratio, margin_edge_ratio, minimum_face_ratio;
    float *out_box;     // [n][5]
    float *out_kps;     // [n][10]
    int *out_found;     // [n]: 0 nothing selected, 1 box only, 3 box + key points
};
int launch_face_select(const SelectParams &p, hipStream_t s);

// FaceAlignment::call on the device (face_alignment.rs:27-141): 5-point similarity to the template + cv::warpAffine
// (INTER_LINEAR, BORDER_CONSTANT 0) of the source frame, or the reference's crop + resize fallback
struct AlignFace {      // per face, filled by the set-up kernel
    double M[6];        // inverse map (dst -> src), as cv::warpAffine computes it
    double scale_x, scale_y; // fallback resize
    int mode;           // 0 warp, 1 crop + resize, < 0 nothing written (status)
    int x0, y0, rw, rh; // fallback ROI
    int area_fast;
};
struct AlignParams {
    const PreImage *imgs; // [n] frames of the batch (device descriptors)
    const float *box;     // [n][5] selected detection
    const float *kps;     // [n][10] its key points
    const int *found;     // [n] selection flags: bit 0 box, bit 1 key points
    float std_lmk[10];    // template (config.rs:46-52)
    int out_w, out_h, n;
    AlignFace *faces;     // [n] scratch
    int *status;          // [n]: 0 aligned, 1 fallback crop, -1 no key points (the reference's call errors), -2 no face,
                          //      -3 fallback ROI outside the frame (Mat::roi error)
    uint8_t *out;         // [n][out_h][out_w][3] u8 BGR
};
int launch_face_align(const AlignParams &p, hipStream_t s);

// ---------------------------------------------------------------- convolution engine (kernels_conv.hip)
// Activations: NHWC bf16.  Weights: [Cout][KH][KW][Cin] bf16 (K contiguous).  f32 accumulate on MFMA.
struct ConvParams {
    const bf16_t *x;      // [B][H][W][Cin]
    const bf16_t *w;      // [Cout][KH*KW*Cin (+ Cin2)]: row pitch = total K
    const bf16_t *x2;     // optional second K segment: a 1x1 conv (stride2, no pad) over [B][H2][W2][Cin2]
    const float *bias2;   // its bias (added to `bias`), or null
    const float *bias;    // [Cout] (BN folded)
    const bf16_t *zero;   // >= 16 bytes of zeros (source of padding taps for the LDS-DMA)
    const bf16_t *res;    // residual [B][RH][RW][Cout] or null; added before relu / raw store
    const float *in_scale; // optional per-INPUT-channel affine + ReLU applied to the im2col operand (1x1 convs
    const float *in_shift; // only): x' = relu(x * in_scale[c] + in_shift[c]) -- the BN+ReLU of the producer unit
    const float *scale2;  // second output: act = relu(v * scale2 + shift2), or null
    const float *shift2;
    bf16_t *y;            // primary output [B][Ho][Wo][ldy] at channel offset y_coff (null: skip)
    bf16_t *y2;           // activated second output [B][Ho][Wo][Cout] or null
    float *yf;            // f32 output [B][Ho][Wo][Cout] (heads) or null
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int H2, W2, Cin2, stride2;
    int ldy, y_coff;      // primary output row pitch (channels) and channel offset (SSH concat)
    int ldx, x_coff;      // input row pitch (channels per pixel, >= Cin) and channel offset: reads a channel slice
    int y_split, y_split_add; // output channels >= y_split land y_split_add further (two destinations, one GEMM)
    int n_valid;          // primary-output channels >= n_valid (zero-padded weight rows) are not stored
    int relu;             // relu on the primary output
    int res_up2;          // residual is half resolution: read at (ho/2, wo/2) (FPN nearest 2x)
    int res_post;         // add the residual AFTER the ReLU (FPN: relu(lateral) + upsampled)
    int force_tile;       // 0 = heuristic, 1 = 128-row tiles, 2 = 256x128 tile (tuning / tests)
    int co_running;       // another chain of the same pass runs concurrently (batch split): affects the tile heuristic
    int head_softmax;     // heads: channels [0,4) are cls logits -> 2-class softmax pairs (a, A+a)
    int k_chunk_major;    // set by launch_conv: K order (chunk, ky, kx) instead of (ky, kx, chunk) (see conv_igemm_kernel)
    // back-to-back pair (OP_B2B beyond stage 1): after this 1x1 conv3 (+ residual -> y = the raw sum), the NEXT unit's conv1 on
    // relu(y * scale2 + shift2) -- or, for the last unit of a stage (y null), on its activated output y2 --:
    // t1 = relu(W1 . act + bias1), [B][H][W][N1], N1 = the next unit's bottleneck width (this Cin; 128 for the stage 1 -> 2 boundary).
    // launch_conv runs the pair in one kernel (pw_b2b_kernel) where that pays and as two launches otherwise: same bits.
    const bf16_t *w1;     // [n1][Cout] (row pitch Cout), or null: no pair
    const float *bias1;
    bf16_t *t1;
    int n1;               // conv1's output channels (128 or 256)
    int *fail;            // device word a kernel with bounded spin waits (kernels_ring.hip) sets when a wave gives up, or null
};
int launch_conv(const ConvParams &p, hipStream_t s);
// wave-specialised loader / consumer ring form of the 128 x 128 implicit-GEMM tile (kernels_ring.hip)
bool conv_ring_supports(const ConvParams &p, bool *kx3);
int launch_conv_ring(const ConvParams &p, hipStream_t s);
// entry i of the list of persistent kernels (name prefix, dynamic LDS every launch of it requests); returns the list length
int persistent_kernel_table(int i, const char **name, size_t *lds_bytes);
// ---- f32 parity mode (kernels_f32.hip): ConvParams with f32 tensors and weights; same field meanings ----
struct ConvF32Params {
    const float *x, *w, *x2;   // [B][H][W][ldx] (channel slice at x_coff), [Cout][ldw], optional [B][H2][W2][Cin2]
    const float *bias, *bias2; // [Cout]; the fused shortcut's bias or null
    const float *res;          // residual [B][RH][RW][Cout] or null
    const float *in_scale, *in_shift, *scale2, *shift2;
    float *y, *y2, *yf;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int H2, W2, Cin2, stride2;
    int ldw;                   // weight row pitch in elements (KH*KW*Cin + Cin2)
    int ldy, y_coff, ldx, x_coff, y_split, y_split_add, n_valid;
    int relu, res_up2, res_post, head_softmax;
};
int launch_conv_f32(const ConvF32Params &p, hipStream_t s);
int launch_conv0_f32(const bf16_t *x4, const float *w, const float *bias, float *y, int B, int H, int W, hipStream_t s);
int launch_maxpool_f32(const float *x, float *y, const float *scale, const float *shift, int B, int H, int W, int C, hipStream_t s);

// back-to-back fusion (stage 1): raw = conv3(x) [+ 1x1 shortcut(x2)] + bias (+ res); t1 = relu(conv1(relu(raw*scale+shift)) + bias1)
struct B2BParams {
    const bf16_t *x, *x2;      // [M][Cin], optional [M][Cin2] (stride-1 shortcut source)
    const bf16_t *w3;          // [256][Cin + Cin2]
    const float *bias3, *bias3b; // conv3 bias, shortcut bias (or null)
    const bf16_t *res;         // [M][256] or null
    const float *scale, *shift; // the unit's post-add affine
    bf16_t *raw;               // [M][256]
    const bf16_t *w1;          // [64][256]
    const float *bias1;
    bf16_t *t1;                // [M][64]
    int B, H, W, Cin, Cin2;
    int force_tile;            // as ConvParams::force_tile (6: persistent form forced, 7 / 1 / 2: never)
};
int launch_conv_b2b_s1(const B2BParams &p, hipStream_t s);
// conv0: 7x7 stride 2 pad 3 on the NHWC4 input, fused bias + ReLU -> [B][H/2][W/2][64]
int launch_conv0(const bf16_t *x4, const bf16_t *w, const float *bias, bf16_t *y, int B, int H,
                 int W, hipStream_t s);
// fused stem: conv0 (7x7/2 + bias + ReLU) -> 3x3/2 max pool -> affine + ReLU, NHWC4 in, [B][H/4][W/4][64] out
// w1 / bias1 / t1 / fused (all or none): the first unit's conv1 (1x1, 64 -> 64, bias + ReLU) computed on the pooled tile and stored to
// t1 when the persistent form runs; *fused tells the caller whether it was (then the conv's own op must not run)
int launch_stem(const bf16_t *x4, const bf16_t *w, const float *bias, const float *scale, const float *shift,
                bf16_t *y, int B, int H, int W, hipStream_t s, const bf16_t *w1 = nullptr, const float *bias1 = nullptr,
                bf16_t *t1 = nullptr, bool *fused = nullptr);
// MobileNet-0.25 helpers: first 3x3/2 conv (3 -> 8 real channels, output padded to Cd) and depthwise 3x3
int launch_first3x3(const bf16_t *x4, const bf16_t *w, const float *bias, bf16_t *y, int B, int H, int W, int Cd,
                    hipStream_t s);
int launch_dwconv3x3(const bf16_t *x, const bf16_t *w, const float *bias, bf16_t *y, int B, int H, int W, int C,
                     int stride, hipStream_t s);
// 3x3 stride 2 pad 1 max pool, NHWC bf16
// optional fused per-channel affine + ReLU on the pooled value (scale/shift may be null)
int launch_maxpool3x3s2(const bf16_t *x, bf16_t *y, const float *scale, const float *shift, int B, int H,
                        int W, int C, hipStream_t s);
// head tensors [B][h][w][32] f32 (cls4 bbox8 lmk20) -> reference NCHW contract (rfd_forward)
int launch_heads_to_nchw(const float *h32, float *cls, float *bbox, float *lmk, int B, int fh,
                         int fw, hipStream_t s);

} // namespace rfd

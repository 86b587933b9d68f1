// detector.hip -- host side of the detection stage: the C++ mirror of the reference's
// `RetinaFaceDetection` (src/pipeline/module/face_detection.rs:19-513) driving the HIP kernels, and
// the C ABI of include/rfd.h on top of it.  (The reference's host language, Rust, has no toolchain
// in this image; INTEGRATION.md shows the `extern "C"` facade that keeps its `call` signature.)
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "network.h"

namespace rfd {

static thread_local char g_err[512] = "";
LaunchNote &launch_note()
{
    static thread_local LaunchNote n;
    return n;
}
bool note_launch(const char *fmt, ...)
{
    LaunchNote &n = launch_note();
    if (!n.dry) return false;
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (!n.names.empty()) n.names += " + ";
    n.names += buf;
    return true;
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }

// ---- anchors: generate_anchors_fpn2 with the production config (face_detection.rs:55-98,
//      generate_anchors.rs:20-39, 61-93, 116-157); f32 arithmetic in the reference's order ----
static void make_base_anchors(float out[kNumLevels][kA][4])
{
    static const float scales[kNumLevels][kA] = {{32.f, 16.f}, {8.f, 4.f}, {2.f, 1.f}};
    const int base_size = 16;
    const float ratio = 1.0f;
    for (int l = 0; l < kNumLevels; ++l) {
        // base anchor [0,0,15,15] -> (w,h,ctr)
        const float b0 = 1.0f - 1.0f, b2 = (float)base_size - 1.0f;
        float w = b2 - b0 + 1.0f, h = b2 - b0 + 1.0f;
        float xc = b0 + 0.5f * (w - 1.0f), yc = b0 + 0.5f * (h - 1.0f);
        // _ratio_enum
        const float ws = roundf(sqrtf(w * h / ratio)), hs = ws * ratio;
        const float r0 = xc - 0.5f * (ws - 1.0f), r1 = yc - 0.5f * (hs - 1.0f);
        const float r2 = xc + 0.5f * (ws - 1.0f), r3 = yc + 0.5f * (hs - 1.0f);
        // _scale_enum
        w = r2 - r0 + 1.0f; h = r3 - r1 + 1.0f;
        xc = r0 + 0.5f * (w - 1.0f); yc = r1 + 0.5f * (h - 1.0f);
        for (int a = 0; a < kA; ++a) {
            const float sw = w * scales[l][a], sh = h * scales[l][a];
            out[l][a][0] = xc - 0.5f * (sw - 1.0f);
            out[l][a][1] = yc - 0.5f * (sh - 1.0f);
            out[l][a][2] = xc + 0.5f * (sw - 1.0f);
            out[l][a][3] = yc + 0.5f * (sh - 1.0f);
        }
    }
}

// ---- _preprocess geometry (face_detection.rs:140-153) + cv::resize's scale bookkeeping ----
static void letterbox(int img_h, int img_w, int size_w, int size_h, PreImage *pi, float *det_scale)
{
    const float im_ratio = (float)img_h / (float)img_w;
    const float model_ratio = (float)size_h / (float)size_w;
    int new_w, new_h;
    if (im_ratio > model_ratio) {
        new_h = size_h;
        new_w = (int)((float)new_h / im_ratio);
    } else {
        new_w = size_w;
        new_h = (int)((float)new_w * im_ratio);
    }
    *det_scale = (float)new_h / (float)img_h;
    pi->h = img_h; pi->w = img_w; pi->new_w = new_w; pi->new_h = new_h; pi->pad = 0;
    pi->scale_x = pi->scale_y = 1.0;
    pi->area_fast = 0;
    if (new_w > 0 && new_h > 0) {
        const double inv_x = (double)new_w / img_w, inv_y = (double)new_h / img_h;
        pi->scale_x = 1.0 / inv_x;
        pi->scale_y = 1.0 / inv_y;
        const int ix = (int)lrint(pi->scale_x), iy = (int)lrint(pi->scale_y);
        pi->area_fast = fabs(pi->scale_x - ix) < 2.220446049250313e-16 &&
                        fabs(pi->scale_y - iy) < 2.220446049250313e-16 && ix == 2 && iy == 2;
    }
}

// ---- RCCL, bound at run time: single-GPU users (and the CPU-side ABI tests) never load the 500 MB library.  Only the
//      handful of entry points the gather needs; types restated from rccl.h (ABI-stable since NCCL 2.0). ----
struct Rccl {
    typedef struct { char internal[RFD_COMM_ID_BYTES]; } UniqueId;
    typedef void *Comm;
    enum { kInt32 = 2 }; // ncclInt32
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *handle = nullptr;
    bool ok = false;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.ok) return RFD_OK;
    const char *env = getenv("RFD_RCCL_LIB");
    void *h = nullptr;
    if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    // a copy some other component of the process already loaded (e.g. the one PyTorch bundles) comes first: one RCCL per process
    static const char *names[] = {"librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; ++pass)
        for (const char *nm : names) {
            h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char *why = dlerror(); // NULL when the last attempt was a NOLOAD probe that found nothing
        set_error("librccl could not be loaded (%s); set RFD_RCCL_LIB", why ? why : "not found on the loader path");
        return RFD_ERR_COMM;
    }
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        set_error("the loaded librccl lacks a required entry point");
        (void)dlclose(h); // do not keep a handle per failed attempt
        return RFD_ERR_COMM;
    }
    r.ok = true;
    g_rccl = r;
    return RFD_OK;
}

#define RFD_RCCL(expr)                                                                                      \
    do {                                                                                                    \
        const int _r = (expr);                                                                              \
        if (_r != 0) {                                                                                      \
            ::rfd::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return RFD_ERR_COMM;                                                                            \
        }                                                                                                   \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return RFD_OK;
        if (p) RFD_HIP(hipFree(p));
        p = nullptr; cap = 0;
        RFD_HIP(hipMalloc(&p, bytes));
        cap = bytes;
        return RFD_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

} // namespace rfd

using namespace rfd;

// The context = the reference's `RetinaFaceDetection` struct (face_detection.rs:19-38) minus the
// Triton client/model config, plus device state.
struct rfd_ctx {
    rfd_config cfg;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    hipEvent_t ev[10] = {};
    float base_anchor[kNumLevels][kA][4];
    int fh[kNumLevels], fw[kNumLevels], level_off[kNumLevels], total_anchors = 0;
    Network net;
    bool net_created = false;
    // device state sized for max_batch_size
    DevBuf staging, imgs, in4, rows, keys, sorted_keys, sorted_boxes, count, det_scale;
    DevBuf nms_kept, nms_state; // chunked NMS (dense crowds): kept-box lists, per-chunk {count, epoch} + the spin_fail word
    int nms_epoch = 0;
    int nms_ticket_sel = 0;  // which of the two chunk-ticket counters the next chunked NMS launch draws from
    bool nms_chunked = true; // RFD_NMS_CHUNKED=0: one workgroup per image always (A/B and fallback)
    // host mirror (page-locked) of the chunked NMS's spin_fail word: copied stream-ordered behind every NMS launch, looked at
    // wherever the host synchronises with the stream (check_nms_flag)
    int *h_nms_flag = nullptr;
    DevBuf out_boxes, out_lmk, out_count, out_total, out_gidx;
    DevBuf scratch[12];
    DevBuf sel_dims, sel_out;
    DevBuf align_faces, align_out, align_status;
    // pinned host ring for per-call descriptors, so enqueueing never blocks on the previous call
    static constexpr int kRing = 4;
    PreImage *pin_imgs[kRing] = {};
    float *pin_scales[kRing] = {};
    hipEvent_t pin_done[kRing] = {};
    int pin_next = 0;
    // pipelined host entry (rfd_submit_batch / rfd_collect_batch): two slots, H2D on its own stream
    struct PipeSlot {
        DevBuf frames, imgs, scale, ob, ol, oc, ot;
        PreImage *pin_imgs = nullptr;
        float *pin_scale = nullptr;
        float *h_ob = nullptr, *h_ol = nullptr;
        int *h_oc = nullptr, *h_ot = nullptr;
        hipEvent_t h2d = nullptr, done = nullptr, post = nullptr;
        int n = 0;
    };
    static constexpr int kPipe = 2;
    static constexpr int kPipeRows = 128; // rows per image the pipelined entry copies back unconditionally
    PipeSlot pipe[kPipe];
    hipStream_t copy_stream = nullptr, d2h_stream = nullptr;
    int pipe_head = 0, pipe_tail = 0, pipe_inflight = 0;
    // cross-call overlap (rfd_detect_batch_device, async = 2): per-parity descriptors and events
    DevBuf ov_imgs[2], ov_scale[2];
    hipEvent_t ov_chain_done[2][2] = {}, ov_post_done[2] = {}, ov_desc = nullptr;
    bool ov_post_valid[2] = {false, false};
    int ov_parity = 0;
    // batch size of the previous call if it ran in the overlap mode, else -1: the part boundary (n+1)/2 and with it the
    // workspace / input slices of the two chains depend on n, and every other entry point runs the network on the
    // caller's stream, so a chain may start under the previous call only when that call had the same shape
    int ov_last_n = -1;
    hipEvent_t ov_resync = nullptr;
    // multi-GPU: RCCL communicator of this rank (rfd_comm_init)
    Rccl::Comm comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    rfd_stats stats;
    float conv_ms = 0.f;
    double conv_flops = 0.0;
    int conv_launches = 0;

    int ensure_network()
    {
        if (net_created) return RFD_OK;
        RFD_TRY(net.create(cfg.backbone, cfg.image_w, cfg.image_h, cfg.max_batch_size, cfg.precision));
        // the ring convolutions' bounded spins report into the same device word as the chunked NMS (check_nms_flag)
        net.d_fail = (int *)nms_state.p + (size_t)cfg.max_batch_size * kNmsChunks * 2;
        net_created = true;
        return RFD_OK;
    }
};

namespace {

int ctx_alloc(rfd_ctx *c)
{
    const size_t B = (size_t)c->cfg.max_batch_size, NA = (size_t)c->total_anchors, MD = (size_t)c->cfg.max_det;
    RFD_TRY(c->imgs.reserve(B * sizeof(PreImage)));
    RFD_TRY(c->rows.reserve(B * NA * kDetRow * sizeof(float)));
    RFD_TRY(c->keys.reserve(B * NA * sizeof(uint64_t)));
    RFD_TRY(c->sorted_keys.reserve(B * NA * sizeof(uint64_t)));
    RFD_TRY(c->sorted_boxes.reserve(B * NA * sizeof(float4)));
    RFD_TRY(c->nms_kept.reserve(B * NA * sizeof(float4)));
    // per-chunk progress words | fault word (+ pad) | the chunk-ticket word (8-byte aligned)
    RFD_TRY(c->nms_state.reserve((B * kNmsChunks * 2 + 4) * sizeof(int)));
    RFD_HIP(hipMemset(c->nms_state.p, 0, (B * kNmsChunks * 2 + 4) * sizeof(int)));
    RFD_HIP(hipDeviceSynchronize()); // the fill runs on the NULL stream, which the context's non-blocking stream is not ordered with
    RFD_TRY(c->count.reserve(B * sizeof(int)));
    RFD_TRY(c->det_scale.reserve(B * sizeof(float)));
    RFD_TRY(c->out_boxes.reserve(B * MD * 5 * sizeof(float)));
    RFD_TRY(c->out_lmk.reserve(B * MD * 10 * sizeof(float)));
    RFD_TRY(c->out_count.reserve(B * sizeof(int)));
    RFD_TRY(c->out_total.reserve(B * sizeof(int)));
    RFD_TRY(c->out_gidx.reserve(B * MD * sizeof(int)));
    return RFD_OK;
}

int check_images(const rfd_ctx *c, const rfd_image *imgs, int n)
{
    RFD_CHECK_ARG(imgs != nullptr, "imgs is null");
    if (n < 1 || n > c->cfg.max_batch_size) {
        set_error("batch of %d frames exceeds max_batch_size %d", n, c->cfg.max_batch_size);
        return RFD_ERR_CAPACITY;
    }
    for (int i = 0; i < n; ++i) {
        RFD_CHECK_ARG(imgs[i].data != nullptr, "frame data is null");
        RFD_CHECK_ARG(imgs[i].height > 0 && imgs[i].width > 0, "frame has a non-positive size");
        RFD_CHECK_ARG(imgs[i].stride >= (ptrdiff_t)imgs[i].width * 3, "frame stride < width*3 (frames must be 3-channel 8-bit)");
    }
    return RFD_OK;
}

// Stage the frames (host or device resident) and their letterbox geometry on the device.
int stage_frames(rfd_ctx *c, const rfd_image *imgs, int n, bool frames_on_device, std::vector<float> &scales)
{
    const int slot = c->pin_next;
    c->pin_next = (c->pin_next + 1) % rfd_ctx::kRing;
    RFD_HIP(hipEventSynchronize(c->pin_done[slot])); // the copy that last used this slot has run
    PreImage *pis = c->pin_imgs[slot];
    scales.resize(n);
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (size_t)imgs[i].height * imgs[i].width * 3;
    if (!frames_on_device) RFD_TRY(c->staging.reserve(total));
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        letterbox(imgs[i].height, imgs[i].width, c->cfg.image_w, c->cfg.image_h, &pis[i], &scales[i]);
        if (pis[i].new_w <= 0 || pis[i].new_h <= 0) { // the reference's cv::resize errors out on an empty dsize (face_detection.rs:156-159)
            set_error("invalid argument: frame %d (%dx%d) letterboxes to an empty %dx%d image", i, imgs[i].width,
                      imgs[i].height, pis[i].new_w, pis[i].new_h);
            return RFD_ERR_INVALID_ARG;
        }
        if (frames_on_device) {
            pis[i].src = imgs[i].data;
            pis[i].stride = (long long)imgs[i].stride;
        } else {
            uint8_t *dst = (uint8_t *)c->staging.p + off;
            const size_t row = (size_t)imgs[i].width * 3;
            RFD_HIP(hipMemcpy2DAsync(dst, row, imgs[i].data, (size_t)imgs[i].stride, row, imgs[i].height,
                                     hipMemcpyHostToDevice, c->stream));
            pis[i].src = dst;
            pis[i].stride = (long long)row;
            off += row * imgs[i].height;
        }
    }
    memcpy(c->pin_scales[slot], scales.data(), n * sizeof(float));
    RFD_HIP(hipMemcpyAsync(c->imgs.p, pis, n * sizeof(PreImage), hipMemcpyHostToDevice, c->stream));
    RFD_HIP(hipMemcpyAsync(c->det_scale.p, c->pin_scales[slot], n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RFD_HIP(hipEventRecord(c->pin_done[slot], c->stream));
    return RFD_OK;
}

void fill_decode_params(const rfd_ctx *c, DecodeParams &p)
{
    memset(&p, 0, sizeof p);
    for (int l = 0; l < kNumLevels; ++l) {
        p.fh[l] = c->fh[l]; p.fw[l] = c->fw[l]; p.stride[l] = kStrides[l]; p.level_off[l] = c->level_off[l];
    }
    memcpy(p.base_anchor, c->base_anchor, sizeof p.base_anchor);
    p.total_anchors = c->total_anchors;
    p.net_h = c->cfg.image_h; p.net_w = c->cfg.image_w;
    p.conf_thr = c->cfg.confidence_threshold;
    p.rows = (float *)c->rows.p;
    p.keys = (uint64_t *)c->keys.p;
    p.count = (int *)c->count.p;
}

// decode -> sort -> NMS on device-resident heads; outputs to device slabs `o*`.
int post_network(rfd_ctx *c, DecodeParams &dp, bool nchw, int n, float *oboxes, float *olmk, int *ocount,
                 int *ototal, int *ogidx, const float *det_scale = nullptr)
{
    RFD_HIP(hipMemsetAsync(c->count.p, 0, n * sizeof(int), c->stream));
    RFD_TRY(launch_decode(dp, n, nchw, c->stream));
    RFD_HIP(hipEventRecord(c->ev[4], c->stream));
    RFD_TRY(launch_sort((uint64_t *)c->keys.p, (const int *)c->count.p, (const float *)c->rows.p,
                        (uint64_t *)c->sorted_keys.p, (float4 *)c->sorted_boxes.p, c->total_anchors, n, c->stream));
    RFD_HIP(hipEventRecord(c->ev[5], c->stream));
    NmsParams np;
    memset(&np, 0, sizeof np);
    np.sorted_keys = (const uint64_t *)c->sorted_keys.p;
    np.sorted_boxes = (const float4 *)c->sorted_boxes.p;
    np.rows = (const float *)c->rows.p;
    np.count = (const int *)c->count.p;
    np.det_scale = det_scale ? det_scale : (const float *)c->det_scale.p;
    np.presorted_n = -1;
    np.total_anchors = c->total_anchors;
    np.max_det = c->cfg.max_det;
    np.iou_thr = c->cfg.iou_threshold;
    np.out_boxes = oboxes; np.out_lmk = olmk; np.out_count = ocount; np.out_total = ototal; np.out_gidx = ogidx;
    if (c->nms_chunked) { // kNmsChunks workgroups per image; images with few candidates are done by the first alone
        np.kept_boxes = (float4 *)c->nms_kept.p;
        np.chunk_state = (int *)c->nms_state.p;
        np.spin_fail = (int *)c->nms_state.p + (size_t)c->cfg.max_batch_size * kNmsChunks * 2;
        np.ticket = (unsigned *)(np.spin_fail + 2);
        np.ticket_sel = c->nms_ticket_sel;
        c->nms_epoch = c->nms_epoch == 0x7fffffff ? 1 : c->nms_epoch + 1;
        np.epoch = c->nms_epoch;
    }
    bool used_chunked = false;
    RFD_TRY(launch_nms(np, n, c->stream, &used_chunked));
    if (used_chunked) c->nms_ticket_sel ^= 1; // that launch leaves its counter at the grid size and zeroes the other one
    RFD_HIP(hipEventRecord(c->ev[6], c->stream));
    // the device word is sticky (the kernel only ever sets it), so a later call's copy cannot hide an earlier give-up
    // (copied whether or not the chunked kernel ran: the ring convolutions of the network pass report into the same word)
    RFD_HIP(hipMemcpyAsync(c->h_nms_flag, (int *)c->nms_state.p + (size_t)c->cfg.max_batch_size * kNmsChunks * 2, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    return RFD_OK;
}

// Call after the host has synchronised with c->stream.  A chunk workgroup of nms_chunked_kernel that gave up waiting for its
// predecessor (bounded spin; forward progress rests on in-order workgroup dispatch, which the hardware does not promise)
// produced a wrong kept set: that is an error, as every failure of the reference's call is an Err (face_detection.rs:498-509),
// never RFD_OK with wrong detections.
int check_nms_flag(rfd_ctx *c)
{
    if (!c->h_nms_flag || *c->h_nms_flag == 0) return RFD_OK;
    *c->h_nms_flag = 0;
    int *flag = (int *)c->nms_state.p + (size_t)c->cfg.max_batch_size * kNmsChunks * 2;
    RFD_HIP(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream));
    set_error("a device-side bounded wait gave up (chunked NMS: a workgroup waiting for its predecessor chunk; ring convolution: a "
              "wave waiting for a ring slot); the detections of the batches since the last synchronisation are invalid "
              "(re-submit them; RFD_NMS_CHUNKED=0 selects the one-workgroup-per-image NMS kernel, RFD_CONV_RING=0 the barrier-per-step convolutions)");
    return RFD_ERR_HIP;
}

int finish_stats(rfd_ctx *c, int n, bool have_pre, bool have_net)
{
    float ms;
    memset(&c->stats, 0, sizeof c->stats);
    auto el = [&](int a, int b) { return hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) == hipSuccess ? ms : 0.f; };
    if (have_pre) { c->stats.ms_h2d = el(0, 1); c->stats.ms_preprocess = el(1, 2); }
    if (have_net) c->stats.ms_network = el(2, 3);
    c->stats.ms_decode = el(3, 4);
    c->stats.ms_sort = el(4, 5);
    c->stats.ms_nms = el(5, 6);
    c->stats.ms_d2h = el(6, 7);
    c->stats.ms_total = el(have_pre ? 0 : (have_net ? 2 : 3), 7);
    std::vector<int> cnt(n), tot(n);
    RFD_HIP(hipMemcpy(cnt.data(), c->count.p, n * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) c->stats.candidates += cnt[i];
    if (c->net_created && c->net.profiling) {
        RFD_TRY(c->net.collect_profile());
        c->conv_ms = 0.f; c->conv_flops = 0.0; c->conv_launches = 0;
        for (size_t i = 0; i < c->net.g.ops.size(); ++i)
            if (c->net.g.ops[i].kind == OP_CONV || c->net.g.ops[i].kind == OP_B2B) {
                c->conv_ms += c->net.op_ms[i];
                c->conv_flops += 2.0 * c->net.g.layer_macs((int)i) * n;
                ++c->conv_launches;
            }
    }
    return RFD_OK;
}

// Cross-call overlap (rfd_detect_batch_device with async = 2; frames and outputs in HBM, frames complete at call time).
// The two parts of the batch run as chains on their OWN streams (not the caller's): preprocess of the part -> network ->
// (side streams joined).  Nothing of call i+1 waits for call i except through the chain's own stream order, so the head
// of chain A of call i+1 overlaps the tail of chain B of call i and the decode / sort / NMS of call i, which stay on the
// caller's stream behind both chains.  Hazards and how they are closed:
//   heads, frame descriptors, det_scale : double-buffered by call parity; a chain of call i+2 first waits for the
//                                          post-processing of call i (ov_post_done) before it touches that parity again
//   workspace slices, network input       : private to a part, protected by the part stream's order
//   output slabs (caller's)               : written by NMS on the caller's stream, i.e. in the caller's own order
int detect_overlapped(rfd_ctx *c, const rfd_image *imgs, int n, rfd_dets *out, hipEvent_t frames_ready = nullptr)
{
    Network &net = c->net;
    RFD_TRY(net.ensure_alt_heads());
    const size_t B = (size_t)c->cfg.max_batch_size;
    if (!c->ov_desc) {
        RFD_HIP(hipEventCreateWithFlags(&c->ov_desc, hipEventDisableTiming));
        RFD_HIP(hipEventCreateWithFlags(&c->ov_resync, hipEventDisableTiming));
        for (int a = 0; a < 2; ++a) {
            RFD_HIP(hipEventCreateWithFlags(&c->ov_post_done[a], hipEventDisableTiming));
            for (int b = 0; b < 2; ++b) RFD_HIP(hipEventCreateWithFlags(&c->ov_chain_done[a][b], hipEventDisableTiming));
            RFD_TRY(c->ov_imgs[a].reserve(B * sizeof(PreImage)));
            RFD_TRY(c->ov_scale[a].reserve(B * sizeof(float)));
        }
    }
    const int par = (c->ov_parity ^= 1);
    const int slot = c->pin_next;
    c->pin_next = (c->pin_next + 1) % rfd_ctx::kRing;
    RFD_HIP(hipEventSynchronize(c->pin_done[slot]));
    PreImage *pis = c->pin_imgs[slot];
    for (int i = 0; i < n; ++i) {
        letterbox(imgs[i].height, imgs[i].width, c->cfg.image_w, c->cfg.image_h, &pis[i], &c->pin_scales[slot][i]);
        if (pis[i].new_w <= 0 || pis[i].new_h <= 0) {
            set_error("invalid argument: frame %d (%dx%d) letterboxes to an empty image", i, imgs[i].width, imgs[i].height);
            return RFD_ERR_INVALID_ARG;
        }
        pis[i].src = imgs[i].data;
        pis[i].stride = (long long)imgs[i].stride;
    }
    hipStream_t st[2] = {net.part_stream[0], net.part_stream[1]};
    if (c->ov_last_n != n) {
        // Different slices than the previous call's chains (another batch size), or the previous call ran on the caller's
        // stream (any other entry point): both chains start behind everything enqueued so far.  The caller's stream has
        // already waited for every earlier chain (post_network of each overlapped call sits behind ov_chain_done).
        RFD_HIP(hipEventRecord(c->ov_resync, c->stream));
        for (int p = 0; p < 2; ++p) RFD_HIP(hipStreamWaitEvent(st[p], c->ov_resync, 0));
    }
    c->ov_last_n = n;
    // (Round 3 also ran each call as ONE whole-batch chain on the stream of its parity, the next call beside it in a second copy of
    //  the workspace: bit-exact, and no faster than the half-batch chains -- 7 721 / 7 432 / 7 474 / 7 534 against 7 439 / 7 642 /
    //  7 486 / 7 421 img/s, profiles/r03_ab_whole_call_chains.jsonl: at 32 images nearly every kernel fills the chip, so the two
    //  chains simply alternate.  Removed.)
    if (frames_ready) // pipelined host entry: the frames of this call are still crossing PCIe on the copy stream
        for (int p = 0; p < 2; ++p) RFD_HIP(hipStreamWaitEvent(st[p], frames_ready, 0));
    if (c->ov_post_valid[par])
        for (int p = 0; p < 2; ++p) RFD_HIP(hipStreamWaitEvent(st[p], c->ov_post_done[par], 0));
    RFD_HIP(hipMemcpyAsync(c->ov_imgs[par].p, pis, n * sizeof(PreImage), hipMemcpyHostToDevice, st[0]));
    RFD_HIP(hipMemcpyAsync(c->ov_scale[par].p, c->pin_scales[slot], n * sizeof(float), hipMemcpyHostToDevice, st[0]));
    RFD_HIP(hipEventRecord(c->pin_done[slot], st[0]));
    RFD_HIP(hipEventRecord(c->ov_desc, st[0]));
    RFD_HIP(hipStreamWaitEvent(st[1], c->ov_desc, 0));
    net.head_parity = par;
    net.co_running = 1;
    const int B0 = (n + 1) / 2;
    const size_t in_px = (size_t)c->cfg.image_h * c->cfg.image_w * 4;
    int status = RFD_OK;
    for (int p = 0; p < 2 && status == RFD_OK; ++p) {
        const int off = p ? B0 : 0, Bp = p ? n - B0 : B0;
        PreParams pp;
        memset(&pp, 0, sizeof pp);
        pp.imgs = (const PreImage *)c->ov_imgs[par].p + off;
        pp.net_h = c->cfg.image_h; pp.net_w = c->cfg.image_w;
        pp.out_nhwc4 = (bf16_t *)net.tensor_ptr(net.g.input) + (size_t)off * in_px;
        status = launch_preprocess(pp, Bp, st[p]);
        if (status == RFD_OK && p == 1 && net.chain_shift_op >= 0 && net.chain_shift_op < (int)net.g.ops.size() &&
            hipStreamWaitEvent(st[1], net.ev_shift, 0) != hipSuccess) status = RFD_ERR_HIP;
        if (status == RFD_OK) status = net.run(Bp, st[p], 0, -1, off, p);
        if (status == RFD_OK && hipEventRecord(c->ov_chain_done[par][p], st[p]) != hipSuccess) status = RFD_ERR_HIP;
    }
    net.co_running = 0;
    if (status != RFD_OK) { net.head_parity = 0; c->ov_last_n = -1; return status; }
    for (int p = 0; p < 2; ++p) RFD_HIP(hipStreamWaitEvent(c->stream, c->ov_chain_done[par][p], 0));
    DecodeParams dp;
    fill_decode_params(c, dp);
    for (int l = 0; l < kNumLevels; ++l) dp.cls[l] = (const float *)net.tensor_ptr(net.g.heads[l]);
    net.head_parity = 0;
    RFD_TRY(post_network(c, dp, false, n, out->boxes, out->landmarks, out->count, out->total, nullptr,
                         (const float *)c->ov_scale[par].p));
    RFD_HIP(hipEventRecord(c->ov_post_done[par], c->stream));
    c->ov_post_valid[par] = true;
    return RFD_OK;
}

int detect_impl(rfd_ctx *c, const rfd_image *imgs, int n, rfd_dets *out, bool on_device, int async, bool frames_on_device)
{
    RFD_CHECK_ARG(c != nullptr, "ctx is null");
    RFD_CHECK_ARG(out && out->boxes && out->landmarks && out->count, "output buffers are null");
    RFD_TRY(check_images(c, imgs, n));
    RFD_TRY(c->ensure_network());
    if (!c->net.weights_ready) { set_error("network weights are not initialised"); return RFD_ERR_STATE; }
    if (async == 2 && on_device && frames_on_device && !c->net.profiling && c->net.multi_stream && c->net.num_parts(n) == 2) {
        RFD_HIP(hipSetDevice(c->cfg.device_id));
        return detect_overlapped(c, imgs, n, out);
    }
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    c->ov_last_n = -1;
    std::vector<float> scales;
    RFD_HIP(hipEventRecord(c->ev[0], c->stream));
    RFD_TRY(stage_frames(c, imgs, n, frames_on_device, scales));
    RFD_HIP(hipEventRecord(c->ev[1], c->stream));
    PreParams pp;
    memset(&pp, 0, sizeof pp);
    pp.imgs = (const PreImage *)c->imgs.p;
    pp.net_h = c->cfg.image_h; pp.net_w = c->cfg.image_w;
    pp.out_nhwc4 = (bf16_t *)c->net.tensor_ptr(c->net.g.input);
    RFD_TRY(launch_preprocess(pp, n, c->stream));
    RFD_HIP(hipEventRecord(c->ev[2], c->stream));
    RFD_TRY(c->net.run_graphed(n, c->stream));
    RFD_HIP(hipEventRecord(c->ev[3], c->stream));
    DecodeParams dp;
    fill_decode_params(c, dp);
    for (int l = 0; l < kNumLevels; ++l) dp.cls[l] = (const float *)c->net.tensor_ptr(c->net.g.heads[l]);
    float *ob = on_device ? out->boxes : (float *)c->out_boxes.p;
    float *ol = on_device ? out->landmarks : (float *)c->out_lmk.p;
    int *oc = on_device ? out->count : (int *)c->out_count.p;
    int *ot = on_device ? out->total : (int *)c->out_total.p;
    RFD_TRY(post_network(c, dp, false, n, ob, ol, oc, ot, nullptr));
    if (!on_device) {
        const size_t MD = (size_t)c->cfg.max_det;
        RFD_HIP(hipMemcpyAsync(out->boxes, ob, n * MD * 5 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RFD_HIP(hipMemcpyAsync(out->landmarks, ol, n * MD * 10 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RFD_HIP(hipMemcpyAsync(out->count, oc, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if (out->total) RFD_HIP(hipMemcpyAsync(out->total, ot, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    RFD_HIP(hipEventRecord(c->ev[7], c->stream));
    if (async && on_device) return RFD_OK;
    RFD_HIP(hipStreamSynchronize(c->stream));
    RFD_TRY(check_nms_flag(c));
    RFD_TRY(finish_stats(c, n, true, true));
    if (!on_device)
        for (int i = 0; i < n; ++i) c->stats.detections += out->total ? out->total[i] : out->count[i];
    return RFD_OK;
}

std::mutex g_nms_mu;
rfd_ctx *g_nms_ctx[16] = {};

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

void rfd_config_default(rfd_config *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->image_w = 640;                 // config.rs:26
    cfg->image_h = 640;
    cfg->max_batch_size = 1;            // config.rs:28
    cfg->confidence_threshold = 0.7f;   // config.rs:29
    cfg->iou_threshold = 0.45f;         // config.rs:30
    cfg->device_id = 0;
    cfg->max_det = 1024;
    cfg->max_src_w = 3840;
    cfg->max_src_h = 2160;
    cfg->backbone = RFD_BACKBONE_R50;
}

int rfd_version(void) { return RFD_VERSION; }
const char *rfd_last_error(void) { return get_error(); }

int rfd_create(const rfd_config *cfg, rfd_ctx **out)
{
    RFD_CHECK_ARG(cfg && out, "cfg/out is null");
    *out = nullptr;
    RFD_CHECK_ARG(cfg->image_w > 0 && cfg->image_h > 0 && cfg->image_w % 32 == 0 && cfg->image_h % 32 == 0,
                  "image_size must be positive multiples of 32");
    RFD_CHECK_ARG(cfg->max_batch_size >= 1, "max_batch_size < 1");
    RFD_CHECK_ARG(cfg->max_det >= 1, "max_det < 1");
    RFD_CHECK_ARG(cfg->precision == RFD_PRECISION_BF16 || cfg->precision == RFD_PRECISION_F32, "precision must be RFD_PRECISION_BF16 or RFD_PRECISION_F32");
    RFD_CHECK_ARG(cfg->precision == RFD_PRECISION_BF16 || cfg->backbone == RFD_BACKBONE_R50, "the f32 parity mode exists for RetinaFace-R50 only");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device is available: librfd_hip has no CPU fallback");
        return RFD_ERR_NO_DEVICE;
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) {
        set_error("device_id %d out of range (%d devices)", cfg->device_id, ndev);
        return RFD_ERR_NO_DEVICE;
    }
    if (hipSetDevice(cfg->device_id) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", cfg->device_id);
        return RFD_ERR_NO_DEVICE;
    }
    rfd_ctx *c = new rfd_ctx();
    c->cfg = *cfg;
    if (const char *e = getenv("RFD_NMS_CHUNKED")) c->nms_chunked = atoi(e) != 0;
    make_base_anchors(c->base_anchor);
    int off = 0;
    for (int l = 0; l < kNumLevels; ++l) {
        c->fh[l] = cfg->image_h / kStrides[l];
        c->fw[l] = cfg->image_w / kStrides[l];
        c->level_off[l] = off;
        off += c->fh[l] * c->fw[l] * kA;
    }
    c->total_anchors = off;
    memset(&c->stats, 0, sizeof c->stats);
    int st = RFD_OK;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) st = RFD_ERR_HIP;
    c->stream = c->own_stream;
    for (int i = 0; i < 10 && st == RFD_OK; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) st = RFD_ERR_HIP;
    for (int i = 0; i < rfd_ctx::kRing && st == RFD_OK; ++i) {
        if (hipHostMalloc((void **)&c->pin_imgs[i], cfg->max_batch_size * sizeof(PreImage)) != hipSuccess ||
            hipHostMalloc((void **)&c->pin_scales[i], cfg->max_batch_size * sizeof(float)) != hipSuccess ||
            hipEventCreate(&c->pin_done[i]) != hipSuccess)
            st = RFD_ERR_HIP;
    }
    if (st == RFD_OK && hipHostMalloc((void **)&c->h_nms_flag, sizeof(int)) != hipSuccess) st = RFD_ERR_HIP;
    if (st == RFD_OK) *c->h_nms_flag = 0;
    if (st == RFD_OK) st = ctx_alloc(c);
    if (st != RFD_OK) {
        if (st == RFD_ERR_HIP && !*get_error()) set_error("HIP stream/event creation failed");
        rfd_destroy(c);
        return st;
    }
    *out = c;
    return RFD_OK;
}

void rfd_destroy(rfd_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device_id);
    (void)hipDeviceSynchronize(); // part / side / copy streams included
    if (c->comm) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    if (c->net_created) c->net.destroy();
    DevBuf *bufs[] = {&c->staging, &c->imgs, &c->in4, &c->rows, &c->keys, &c->sorted_keys, &c->sorted_boxes,
                      &c->count, &c->det_scale, &c->out_boxes, &c->out_lmk, &c->out_count, &c->out_total,
                      &c->out_gidx, &c->nms_kept, &c->nms_state};
    for (DevBuf *b : bufs) b->release();
    for (DevBuf &b : c->scratch) b.release();
    c->sel_dims.release(); c->sel_out.release();
    c->align_faces.release(); c->align_out.release(); c->align_status.release();
    for (int i = 0; i < 10; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < rfd_ctx::kRing; ++i) {
        if (c->pin_imgs[i]) (void)hipHostFree(c->pin_imgs[i]);
        if (c->pin_scales[i]) (void)hipHostFree(c->pin_scales[i]);
        if (c->pin_done[i]) (void)hipEventDestroy(c->pin_done[i]);
    }
    if (c->h_nms_flag) (void)hipHostFree(c->h_nms_flag);
    for (rfd_ctx::PipeSlot &ps : c->pipe) {
        DevBuf *pb[] = {&ps.frames, &ps.imgs, &ps.scale, &ps.ob, &ps.ol, &ps.oc, &ps.ot};
        for (DevBuf *b : pb) b->release();
        void *hp[] = {ps.pin_imgs, ps.pin_scale, ps.h_ob, ps.h_ol, ps.h_oc, ps.h_ot};
        for (void *h : hp)
            if (h) (void)hipHostFree(h);
        if (ps.h2d) (void)hipEventDestroy(ps.h2d);
        if (ps.done) (void)hipEventDestroy(ps.done);
        if (ps.post) (void)hipEventDestroy(ps.post);
    }
    for (int a = 0; a < 2; ++a) {
        c->ov_imgs[a].release(); c->ov_scale[a].release();
        if (c->ov_post_done[a]) (void)hipEventDestroy(c->ov_post_done[a]);
        for (int b = 0; b < 2; ++b)
            if (c->ov_chain_done[a][b]) (void)hipEventDestroy(c->ov_chain_done[a][b]);
    }
    if (c->ov_desc) (void)hipEventDestroy(c->ov_desc);
    if (c->ov_resync) (void)hipEventDestroy(c->ov_resync);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int rfd_get_config(const rfd_ctx *c, rfd_config *cfg)
{
    RFD_CHECK_ARG(c && cfg, "null argument");
    *cfg = c->cfg;
    return RFD_OK;
}

int rfd_get_stats(rfd_ctx *c, rfd_stats *stats)
{
    RFD_CHECK_ARG(c && stats, "null argument");
    *stats = c->stats;
    return RFD_OK;
}

int rfd_set_stream(rfd_ctx *c, void *hip_stream)
{
    RFD_CHECK_ARG(c, "ctx is null");
    RFD_HIP(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RFD_OK;
}

int rfd_set_thresholds(rfd_ctx *c, float confidence_threshold, float iou_threshold)
{
    RFD_CHECK_ARG(c, "ctx is null");
    c->cfg.confidence_threshold = confidence_threshold;
    c->cfg.iou_threshold = iou_threshold;
    return RFD_OK;
}

int rfd_set_profiling(rfd_ctx *c, int enable)
{
    RFD_CHECK_ARG(c, "ctx is null");
    RFD_TRY(c->ensure_network());
    if (enable && c->net.precision != 0) { set_error("per-op profiling is not available in the f32 parity mode"); return RFD_ERR_STATE; }
    c->net.profiling = enable != 0;
    return RFD_OK;
}

int rfd_get_conv_profile(rfd_ctx *c, float *ms_conv, double *flops_conv, int *launches)
{
    RFD_CHECK_ARG(c, "ctx is null");
    if (ms_conv) *ms_conv = c->conv_ms;
    if (flops_conv) *flops_conv = c->conv_flops;
    if (launches) *launches = c->conv_launches;
    return RFD_OK;
}

int rfd_get_op_profile(rfd_ctx *c, float *ms, int cap)
{
    RFD_CHECK_ARG(c && ms, "null argument");
    if (!c->net_created) { set_error("no network"); return RFD_ERR_STATE; }
    const int n = (int)c->net.op_ms.size();
    for (int i = 0; i < n && i < cap; ++i) ms[i] = c->net.op_ms[i];
    return n;
}

// ---- graph description (host only; usable without a GPU) ----
int rfd_graph_create(int backbone, int image_w, int image_h, rfd_graph **out)
{
    RFD_CHECK_ARG(out, "out is null");
    Graph *g = new Graph();
    const int st = g->build(backbone, image_w, image_h);
    if (st != RFD_OK) { delete g; *out = nullptr; return st; }
    *out = reinterpret_cast<rfd_graph *>(g);
    return RFD_OK;
}
void rfd_graph_destroy(rfd_graph *g) { delete reinterpret_cast<Graph *>(g); }
int rfd_graph_counts(const rfd_graph *gg, int *layers, int *ops, int *tensors, int *buffers)
{
    RFD_CHECK_ARG(gg, "graph is null");
    const Graph *g = reinterpret_cast<const Graph *>(gg);
    if (layers) *layers = (int)g->layers.size();
    if (ops) *ops = (int)g->ops.size();
    if (tensors) *tensors = (int)g->tensors.size();
    if (buffers) *buffers = (int)g->buffer_bytes_per_image.size();
    return RFD_OK;
}
int rfd_graph_layer(const rfd_graph *gg, int idx, rfd_layer_desc *d)
{
    RFD_CHECK_ARG(gg && d, "null argument");
    const Graph *g = reinterpret_cast<const Graph *>(gg);
    RFD_CHECK_ARG(idx >= 0 && idx < (int)g->layers.size(), "layer index out of range");
    const Layer &L = g->layers[idx];
    memset(d, 0, sizeof *d);
    snprintf(d->name, sizeof d->name, "%s", L.name.c_str());
    d->cin = L.cin; d->cout = L.cout; d->kh = L.kh; d->kw = L.kw; d->stride = L.stride; d->pad = L.pad;
    d->has_affine = L.has_affine;
    d->kind = L.kind;
    return RFD_OK;
}
int rfd_graph_op(const rfd_graph *gg, int idx, rfd_op_desc *d)
{
    RFD_CHECK_ARG(gg && d, "null argument");
    const Graph *g = reinterpret_cast<const Graph *>(gg);
    RFD_CHECK_ARG(idx >= 0 && idx < (int)g->ops.size(), "op index out of range");
    const Op &o = g->ops[idx];
    memset(d, 0, sizeof *d);
    d->kind = o.kind; d->layer = o.layer; d->in = o.in; d->out = o.out; d->out2 = o.out2; d->outf = o.outf;
    d->res = o.res; d->relu = o.relu; d->res_up2 = o.res_up2; d->res_post = o.res_post;
    d->head_softmax = o.head_softmax; d->y_coff = o.y_coff;
    d->in2 = o.in2; d->layer2 = o.layer2; d->in_affine = o.in_affine;
    d->layer_n2 = o.layer_n2; d->x_coff = o.x_coff; d->y_split = o.y_split; d->y_split_add = o.y_split_add;
    d->n_valid = o.n_valid; d->layer_b = o.layer_b; d->out_b = o.out_b; d->branch = o.branch;
    d->macs = g->layer_macs(idx);
    return RFD_OK;
}
int rfd_graph_tensor(const rfd_graph *gg, int idx, rfd_tensor_desc *d)
{
    RFD_CHECK_ARG(gg && d, "null argument");
    const Graph *g = reinterpret_cast<const Graph *>(gg);
    RFD_CHECK_ARG(idx >= 0 && idx < (int)g->tensors.size(), "tensor index out of range");
    const TensorDesc &t = g->tensors[idx];
    memset(d, 0, sizeof *d);
    d->channels = t.C; d->channels_logical = t.C_logical; d->height = t.H; d->width = t.W; d->is_f32 = t.is_f32;
    d->buffer = t.buffer;
    d->is_input = idx == g->input;
    for (int l = 0; l < 3; ++l)
        if (g->heads[l] == idx) d->head_level = l + 1;
    return RFD_OK;
}
double rfd_graph_macs(const rfd_graph *gg)
{
    return gg ? reinterpret_cast<const Graph *>(gg)->macs_per_image() : 0.0;
}
double rfd_graph_workspace_bytes(const rfd_graph *gg)
{
    if (!gg) return 0.0;
    double s = 0;
    for (size_t b : reinterpret_cast<const Graph *>(gg)->buffer_bytes_per_image) s += (double)b;
    return s;
}

// ---- test hooks: raw tensor access and partial execution of the network ----
int rfd_debug_tensor_io(rfd_ctx *c, int tensor_id, int n, void *host, int write)
{
    RFD_CHECK_ARG(c && host, "null argument");
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_TRY(c->ensure_network());
    RFD_CHECK_ARG(tensor_id >= 0 && tensor_id < (int)c->net.g.tensors.size(), "tensor id out of range");
    RFD_CHECK_ARG(n >= 1 && n <= c->cfg.max_batch_size, "batch out of range");
    const size_t bytes = c->net.g.tensors[tensor_id].bytes_per_image() * (size_t)n;
    c->ov_last_n = -1;
    if (c->net.precision != 0 && !c->net.g.tensors[tensor_id].is_f32 && tensor_id != c->net.g.input) {
        set_error("f32 parity mode: intermediate tensors are f32, not the bf16 layout this hook transfers");
        return RFD_ERR_STATE;
    }
    if (write) RFD_HIP(hipMemcpyAsync(c->net.tensor_ptr(tensor_id), host, bytes, hipMemcpyHostToDevice, c->stream));
    else RFD_HIP(hipMemcpyAsync(host, c->net.tensor_ptr(tensor_id), bytes, hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream));
    return RFD_OK;
}
int rfd_debug_set_conv_tile(rfd_ctx *c, int tile)
{
    RFD_CHECK_ARG(c && tile >= 0 && tile <= 31, "bad argument");
    RFD_TRY(c->ensure_network());
    c->net.force_tile = tile;
    return RFD_OK;
}
int rfd_debug_set_concurrency(rfd_ctx *c, int multi_stream, int split_min_part, int split_max_parts, int use_graph)
{
    RFD_CHECK_ARG(c, "ctx is null");
    RFD_TRY(c->ensure_network());
    RFD_HIP(hipStreamSynchronize(c->stream));
    c->net.multi_stream = multi_stream != 0;
    c->net.split_min_part = split_min_part;
    c->net.split_max_parts = split_max_parts;
    c->net.tuned = false; // the stream choice depends on the number of parts
    c->net.use_graph = use_graph != 0;
    for (hipGraphExec_t &ge : c->net.graph_exec)
        if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
    return RFD_OK;
}
int rfd_debug_poke_nms_flag(rfd_ctx *c, int value)
{
    RFD_CHECK_ARG(c, "ctx is null");
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    int *flag = (int *)c->nms_state.p + (size_t)c->cfg.max_batch_size * kNmsChunks * 2;
    RFD_HIP(hipMemcpyAsync(flag, &value, sizeof(int), hipMemcpyHostToDevice, c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream));
    return RFD_OK;
}
int rfd_debug_persistent_kernel(int i, const char **name, size_t *lds_bytes) { return rfd::persistent_kernel_table(i, name, lds_bytes); }
int rfd_debug_run_ops(rfd_ctx *c, int n, int first_op, int last_op)
{
    RFD_CHECK_ARG(c, "ctx is null");
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_TRY(c->ensure_network());
    c->ov_last_n = -1;
    RFD_TRY(c->net.run(n, c->stream, first_op, last_op));
    RFD_HIP(hipMemcpyAsync(c->h_nms_flag, c->net.d_fail, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream));
    if (c->net.profiling) RFD_TRY(c->net.collect_profile());
    return check_nms_flag(c);
}

int rfd_debug_op_kernels(rfd_ctx *c, int n, int op, int co_running, char *names, int cap)
{
    RFD_CHECK_ARG(c && names && cap > 0, "null argument");
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_TRY(c->ensure_network());
    RFD_CHECK_ARG(n >= 1 && n <= c->cfg.max_batch_size && op >= 0 && op < (int)c->net.g.ops.size(), "batch or op out of range");
    if (c->net.precision != 0) { set_error("kernel-choice introspection covers the bf16 path"); return RFD_ERR_STATE; }
    LaunchNote &note = launch_note();
    note.names.clear();
    note.dry = true;                       // every launch path records its kernel and returns without launching
    const int saved = c->net.co_running;
    const bool prof = c->net.profiling;
    c->net.co_running = co_running != 0;   // as for a chain of a split pass (batches >= 16 run as two chains of n / 2 images)
    c->net.profiling = false;
    // the stem and the conv behind it may run as ONE launch when a pass runs them back to back (Network::run's peephole): ask for the
    // pair, and report the conv as fused when the pair took a single launch
    const auto &ops = c->net.g.ops;
    const bool is_stem = ops[op].kind == OP_STEM && op + 1 < (int)ops.size(), after_stem = op > 0 && ops[op - 1].kind == OP_STEM;
    const int st = c->net.run(n, c->stream, after_stem ? op - 1 : op, is_stem ? op + 1 : op);
    c->net.co_running = saved;
    c->net.profiling = prof;
    note.dry = false;
    RFD_TRY(st);
    std::string out = note.names;
    if (is_stem || after_stem) {
        const size_t cut = out.find(" + ");
        const bool fused = cut == std::string::npos; // one launch for the two ops
        if (is_stem) out = fused ? out : out.substr(0, cut);
        else out = fused ? "(fused into " + out + ")" : out.substr(cut + 3);
    }
    snprintf(names, (size_t)cap, "%s", out.c_str());
    return RFD_OK;
}

// ---- weights ----
int rfd_init_synthetic_weights(rfd_ctx *c, uint64_t seed)
{
    RFD_CHECK_ARG(c, "ctx is null");
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_TRY(c->ensure_network());
    return c->net.init_synthetic(seed, c->stream);
}
int rfd_num_layers(const rfd_ctx *c)
{
    if (!c || !c->net_created) return 0;
    return (int)c->net.g.layers.size();
}
int rfd_get_layer_weights(rfd_ctx *c, int idx, float *weights, float *bias)
{
    RFD_CHECK_ARG(c, "ctx is null");
    RFD_TRY(c->ensure_network());
    return c->net.get_layer(idx, weights, bias, c->stream);
}
int rfd_set_layer_weights(rfd_ctx *c, int idx, const float *weights, const float *bias)
{
    RFD_CHECK_ARG(c && weights, "null argument");
    RFD_TRY(c->ensure_network());
    RFD_TRY(c->net.set_layer(idx, weights, bias, c->stream));
    c->net.weights_ready = true;
    return RFD_OK;
}
int rfd_get_layer_affine(rfd_ctx *c, int idx, float *scale, float *shift)
{
    RFD_CHECK_ARG(c && scale && shift, "null argument");
    RFD_TRY(c->ensure_network());
    return c->net.get_affine(idx, scale, shift, c->stream);
}
int rfd_set_layer_affine(rfd_ctx *c, int idx, const float *scale, const float *shift)
{
    RFD_CHECK_ARG(c && scale && shift, "null argument");
    RFD_TRY(c->ensure_network());
    return c->net.set_affine(idx, scale, shift, c->stream);
}

namespace {
struct LayerRecord { char name[64]; int32_t cin, cout, kh, kw, stride, pad, kind, has_affine; };
}

int rfd_save_weights(rfd_ctx *c, const char *path)
{
    RFD_CHECK_ARG(c && path, "null argument");
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_TRY(c->ensure_network());
    if (!c->net.weights_ready) { set_error("network weights are not initialised"); return RFD_ERR_STATE; }
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot open %s for writing", path); return RFD_ERR_IO; }
    const uint32_t hdr[3] = {1u, (uint32_t)c->cfg.backbone, (uint32_t)c->net.g.layers.size()};
    bool ok = fwrite("RFDW", 1, 4, f) == 4 && fwrite(hdr, 4, 3, f) == 3;
    for (size_t i = 0; ok && i < c->net.g.layers.size(); ++i) {
        const Layer &L = c->net.g.layers[i];
        LayerRecord rec;
        memset(&rec, 0, sizeof rec);
        snprintf(rec.name, sizeof rec.name, "%s", L.name.c_str());
        rec.cin = L.cin; rec.cout = L.cout; rec.kh = L.kh; rec.kw = L.kw; rec.stride = L.stride; rec.pad = L.pad;
        rec.kind = L.kind; rec.has_affine = L.has_affine;
        const size_t nw = (size_t)L.cout * L.kh * L.kw * L.cin;
        std::vector<float> w(nw), b(L.cout), sc(L.cout), sh(L.cout);
        const int st = c->net.get_layer((int)i, w.data(), b.data(), c->stream);
        if (st != RFD_OK) { fclose(f); return st; }
        ok = fwrite(&rec, sizeof rec, 1, f) == 1 && fwrite(w.data(), 4, nw, f) == nw && fwrite(b.data(), 4, L.cout, f) == (size_t)L.cout;
        if (ok && L.has_affine) {
            const int sa = c->net.get_affine((int)i, sc.data(), sh.data(), c->stream);
            if (sa != RFD_OK) { fclose(f); return sa; }
            ok = fwrite(sc.data(), 4, L.cout, f) == (size_t)L.cout && fwrite(sh.data(), 4, L.cout, f) == (size_t)L.cout;
        }
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("short write to %s", path); return RFD_ERR_IO; }
    return RFD_OK;
}

int rfd_load_weights(rfd_ctx *c, const char *path)
{
    RFD_CHECK_ARG(c && path, "null argument");
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_TRY(c->ensure_network());
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return RFD_ERR_IO; }
    char magic[4];
    uint32_t hdr[3];
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "RFDW", 4) != 0 || fread(hdr, 4, 3, f) != 3 || hdr[0] != 1u) {
        fclose(f);
        set_error("%s is not an RFDW version-1 weight file", path);
        return RFD_ERR_IO;
    }
    if (hdr[1] != (uint32_t)c->cfg.backbone || hdr[2] != (uint32_t)c->net.g.layers.size()) {
        fclose(f);
        set_error("%s holds backbone %u with %u layers; the context expects backbone %d with %zu layers", path, hdr[1], hdr[2],
                  c->cfg.backbone, c->net.g.layers.size());
        return RFD_ERR_INVALID_ARG;
    }
    for (size_t i = 0; i < c->net.g.layers.size(); ++i) {
        const Layer &L = c->net.g.layers[i];
        LayerRecord rec;
        if (fread(&rec, sizeof rec, 1, f) != 1) { fclose(f); set_error("%s is truncated (layer %zu)", path, i); return RFD_ERR_IO; }
        rec.name[63] = 0;
        if (rec.cin != L.cin || rec.cout != L.cout || rec.kh != L.kh || rec.kw != L.kw || rec.stride != L.stride ||
            rec.kind != L.kind || rec.has_affine != L.has_affine) {
            fclose(f);
            set_error("%s: layer %zu (%s) does not match the graph's %s", path, i, rec.name, L.name.c_str());
            return RFD_ERR_INVALID_ARG;
        }
        const size_t nw = (size_t)L.cout * L.kh * L.kw * L.cin;
        std::vector<float> w(nw), b(L.cout), sc(L.cout), sh(L.cout);
        bool ok = fread(w.data(), 4, nw, f) == nw && fread(b.data(), 4, L.cout, f) == (size_t)L.cout;
        if (ok && L.has_affine) ok = fread(sc.data(), 4, L.cout, f) == (size_t)L.cout && fread(sh.data(), 4, L.cout, f) == (size_t)L.cout;
        if (!ok) { fclose(f); set_error("%s is truncated (layer %zu)", path, i); return RFD_ERR_IO; }
        int st = c->net.set_layer((int)i, w.data(), b.data(), c->stream);
        if (st == RFD_OK && L.has_affine) st = c->net.set_affine((int)i, sc.data(), sh.data(), c->stream);
        if (st != RFD_OK) { fclose(f); return st; }
    }
    fclose(f);
    c->net.weights_ready = true;
    return RFD_OK;
}

// ---- hot path ----
int rfd_detect_batch(rfd_ctx *c, const rfd_image *imgs, int n, rfd_dets *out)
{
    return detect_impl(c, imgs, n, out, false, 0, false);
}
int rfd_detect_batch_device(rfd_ctx *c, const rfd_image *imgs, int n, rfd_dets *out, int async)
{
    RFD_CHECK_ARG(out && out->total, "out->total must be a device buffer for the device entry point");
    return detect_impl(c, imgs, n, out, true, async, true);
}
int rfd_sync(rfd_ctx *c)
{
    RFD_CHECK_ARG(c, "ctx is null");
    RFD_HIP(hipStreamSynchronize(c->stream));
    return check_nms_flag(c);
}

// ---- multi-GPU gather ----
int rfd_comm_get_unique_id(void *id)
{
    RFD_CHECK_ARG(id != nullptr, "id is null");
    RFD_TRY(rccl_load());
    Rccl::UniqueId u;
    RFD_RCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return RFD_OK;
}

int rfd_comm_init(rfd_ctx *c, const void *unique_id, int rank, int world)
{
    RFD_CHECK_ARG(c && unique_id, "null argument");
    RFD_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    if (c->comm) { set_error("the context already has a communicator: call rfd_comm_destroy first"); return RFD_ERR_STATE; }
    RFD_TRY(rccl_load());
    RFD_HIP(hipSetDevice(c->cfg.device_id)); // the communicator binds to the calling thread's current device
    Rccl::UniqueId u;
    memcpy(&u, unique_id, sizeof u);
    Rccl::Comm comm = nullptr;
    RFD_RCCL(g_rccl.CommInitRank(&comm, world, u, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return RFD_OK;
}

int rfd_comm_info(const rfd_ctx *c, int *rank, int *world)
{
    RFD_CHECK_ARG(c, "ctx is null");
    if (rank) *rank = c->comm ? c->comm_rank : 0;
    if (world) *world = c->comm ? c->comm_world : 0;
    return RFD_OK;
}

int rfd_gather_detections(rfd_ctx *c, const rfd_dets *local, int n_local, rfd_dets *all)
{
    RFD_CHECK_ARG(c && local && all, "null argument");
    RFD_CHECK_ARG(local->boxes && local->landmarks && local->count && all->boxes && all->landmarks && all->count,
                  "slab pointers are null");
    RFD_CHECK_ARG((local->total != nullptr) == (all->total != nullptr), "total must be given in both slabs or in neither");
    if (!c->comm) { set_error("no communicator: call rfd_comm_init first"); return RFD_ERR_STATE; }
    if (n_local < 1 || n_local > c->cfg.max_batch_size) { set_error("n_local %d exceeds max_batch_size %d", n_local, c->cfg.max_batch_size); return RFD_ERR_CAPACITY; }
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    const size_t MD = (size_t)c->cfg.max_det, n = (size_t)n_local;
    // four arrays, one fused collective; everything moves as 32-bit words (floats are not interpreted)
    RFD_RCCL(g_rccl.GroupStart());
    int r = g_rccl.AllGather(local->boxes, all->boxes, n * MD * 5, Rccl::kInt32, c->comm, c->stream);
    if (r == 0) r = g_rccl.AllGather(local->landmarks, all->landmarks, n * MD * 10, Rccl::kInt32, c->comm, c->stream);
    if (r == 0) r = g_rccl.AllGather(local->count, all->count, n, Rccl::kInt32, c->comm, c->stream);
    if (r == 0 && local->total) r = g_rccl.AllGather(local->total, all->total, n, Rccl::kInt32, c->comm, c->stream);
    const int e = g_rccl.GroupEnd();
    if (r != 0 || e != 0) {
        set_error("ncclAllGather of the detection slabs failed: %s", g_rccl.GetErrorString(r != 0 ? r : e));
        return RFD_ERR_COMM;
    }
    return RFD_OK;
}

int rfd_comm_destroy(rfd_ctx *c)
{
    RFD_CHECK_ARG(c, "ctx is null");
    if (!c->comm) return RFD_OK;
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_HIP(hipStreamSynchronize(c->stream));
    RFD_RCCL(g_rccl.CommDestroy(c->comm));
    c->comm = nullptr; c->comm_world = 0; c->comm_rank = 0;
    return RFD_OK;
}

// ---- pipelined host entry ----
int rfd_host_alloc(size_t bytes, void **ptr)
{
    RFD_CHECK_ARG(ptr && bytes > 0, "bad argument");
    RFD_HIP(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return RFD_OK;
}
int rfd_host_free(void *ptr)
{
    if (ptr) RFD_HIP(hipHostFree(ptr));
    return RFD_OK;
}

int rfd_submit_batch(rfd_ctx *c, const rfd_image *imgs, int n)
{
    RFD_CHECK_ARG(c != nullptr, "ctx is null");
    RFD_TRY(check_images(c, imgs, n));
    if (c->pipe_inflight >= rfd_ctx::kPipe) {
        set_error("%d batches are already in flight: call rfd_collect_batch first", c->pipe_inflight);
        return RFD_ERR_STATE;
    }
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_TRY(c->ensure_network());
    const size_t B = (size_t)c->cfg.max_batch_size, MD = (size_t)c->cfg.max_det;
    rfd_ctx::PipeSlot &ps = c->pipe[c->pipe_head];
    if (!c->copy_stream) RFD_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!ps.done) { // first use of this slot
        RFD_HIP(hipEventCreateWithFlags(&ps.h2d, hipEventDisableTiming));
        RFD_HIP(hipEventCreateWithFlags(&ps.done, hipEventDisableTiming));
        RFD_HIP(hipHostMalloc((void **)&ps.pin_imgs, B * sizeof(PreImage), hipHostMallocDefault));
        RFD_HIP(hipHostMalloc((void **)&ps.pin_scale, B * sizeof(float), hipHostMallocDefault));
        RFD_HIP(hipHostMalloc((void **)&ps.h_ob, B * MD * 5 * sizeof(float), hipHostMallocDefault));
        RFD_HIP(hipHostMalloc((void **)&ps.h_ol, B * MD * 10 * sizeof(float), hipHostMallocDefault));
        RFD_HIP(hipHostMalloc((void **)&ps.h_oc, B * sizeof(int), hipHostMallocDefault));
        RFD_HIP(hipHostMalloc((void **)&ps.h_ot, B * sizeof(int), hipHostMallocDefault));
        RFD_TRY(ps.imgs.reserve(B * sizeof(PreImage)));
        RFD_TRY(ps.scale.reserve(B * sizeof(float)));
        RFD_TRY(ps.ob.reserve(B * MD * 5 * sizeof(float)));
        RFD_TRY(ps.ol.reserve(B * MD * 10 * sizeof(float)));
        RFD_TRY(ps.oc.reserve(B * sizeof(int)));
        RFD_TRY(ps.ot.reserve(B * sizeof(int)));
    }
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (size_t)imgs[i].height * imgs[i].width * 3;
    if (total > ps.frames.cap) { // growing the frame buffer frees the old one: nothing of this slot is in flight (it was collected)
        RFD_TRY(ps.frames.reserve(total + total / 4));
    }
    // copy stream: frames + descriptors of THIS batch while the main stream still computes the previous one
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        letterbox(imgs[i].height, imgs[i].width, c->cfg.image_w, c->cfg.image_h, &ps.pin_imgs[i], &ps.pin_scale[i]);
        if (ps.pin_imgs[i].new_w <= 0 || ps.pin_imgs[i].new_h <= 0) {
            set_error("invalid argument: frame %d (%dx%d) letterboxes to an empty image", i, imgs[i].width, imgs[i].height);
            return RFD_ERR_INVALID_ARG;
        }
        ps.pin_imgs[i].src = (uint8_t *)ps.frames.p + off;
        ps.pin_imgs[i].stride = (long long)imgs[i].width * 3;
        off += (size_t)imgs[i].width * 3 * imgs[i].height;
    }
    // Frames that lie back to back in host memory with tight rows (a decoder writing into one rfd_host_alloc block) travel as
    // ONE copy per run: every copy command on the copy stream costs the compute streams a little (the same 39 MB as 32 commands
    // measured 0.1 ms per batch slower than as one, round 3).
    for (int i = 0; i < n;) {
        size_t bytes = (size_t)imgs[i].width * 3 * imgs[i].height;
        int j = i + 1;
        if (imgs[i].stride == (ptrdiff_t)imgs[i].width * 3) {
            while (j < n && imgs[j].stride == (ptrdiff_t)imgs[j].width * 3 && imgs[j].data == imgs[i].data + bytes) {
                bytes += (size_t)imgs[j].width * 3 * imgs[j].height;
                ++j;
            }
            RFD_HIP(hipMemcpyAsync((void *)ps.pin_imgs[i].src, imgs[i].data, bytes, hipMemcpyHostToDevice, c->copy_stream));
        } else {
            const size_t row = (size_t)imgs[i].width * 3;
            RFD_HIP(hipMemcpy2DAsync((void *)ps.pin_imgs[i].src, row, imgs[i].data, (size_t)imgs[i].stride, row, imgs[i].height,
                                     hipMemcpyHostToDevice, c->copy_stream));
        }
        i = j;
    }
    RFD_HIP(hipMemcpyAsync(ps.imgs.p, ps.pin_imgs, n * sizeof(PreImage), hipMemcpyHostToDevice, c->copy_stream));
    RFD_HIP(hipMemcpyAsync(ps.scale.p, ps.pin_scale, n * sizeof(float), hipMemcpyHostToDevice, c->copy_stream));
    RFD_HIP(hipEventRecord(ps.h2d, c->copy_stream));
    if (!c->net.profiling && c->net.multi_stream && c->net.num_parts(n) == 2 && c->net.weights_ready) {
        // Round 3: the same two-chain, cross-call-overlapped pass rfd_detect_batch_device(async = 2) runs -- the chains of this
        // batch wait for its frames (h2d) and otherwise only for their own stream order, so they start under the tail and the
        // decode / sort / NMS of the previous batch.  (Until round 2 the whole pass sat on the caller's stream behind the previous
        // batch, D2H included: 7.0 k against 7.5 k img/s device-resident.)
        std::vector<rfd_image> dev(n);
        for (int i = 0; i < n; ++i) {
            dev[i].data = (const uint8_t *)ps.pin_imgs[i].src; dev[i].height = imgs[i].height; dev[i].width = imgs[i].width;
            dev[i].stride = (ptrdiff_t)imgs[i].width * 3;
        }
        rfd_dets dd;
        dd.boxes = (float *)ps.ob.p; dd.landmarks = (float *)ps.ol.p; dd.count = (int *)ps.oc.p; dd.total = (int *)ps.ot.p;
        RFD_TRY(detect_overlapped(c, dev.data(), n, &dd, ps.h2d));
    } else {
        // main stream: the whole hot path of this batch, behind the previous batch
        c->ov_last_n = -1;
        RFD_HIP(hipStreamWaitEvent(c->stream, ps.h2d, 0));
        PreParams pp;
        memset(&pp, 0, sizeof pp);
        pp.imgs = (const PreImage *)ps.imgs.p;
        pp.net_h = c->cfg.image_h; pp.net_w = c->cfg.image_w;
        pp.out_nhwc4 = (bf16_t *)c->net.tensor_ptr(c->net.g.input);
        RFD_TRY(launch_preprocess(pp, n, c->stream));
        RFD_TRY(c->net.run_graphed(n, c->stream));
        DecodeParams dp;
        fill_decode_params(c, dp);
        for (int l = 0; l < kNumLevels; ++l) dp.cls[l] = (const float *)c->net.tensor_ptr(c->net.g.heads[l]);
        RFD_TRY(post_network(c, dp, false, n, (float *)ps.ob.p, (float *)ps.ol.p, (int *)ps.oc.p, (int *)ps.ot.p, nullptr,
                             (const float *)ps.scale.p));
    }
    // detections go back on their own stream: neither the next batch's compute (caller's stream) nor its frames (copy stream,
    // enqueued earlier than this batch's NMS finishes) queue behind them
    if (!c->d2h_stream) RFD_HIP(hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
    if (!ps.post) RFD_HIP(hipEventCreateWithFlags(&ps.post, hipEventDisableTiming));
    RFD_HIP(hipEventRecord(ps.post, c->stream));
    RFD_HIP(hipStreamWaitEvent(c->d2h_stream, ps.post, 0));
    // the first kPipeRows rows of every image now (two strided copies); an image that kept more has the rest fetched by
    // rfd_collect_batch -- dense crowds only: the slabs hold max_det rows per image, a typical frame uses a few dozen
    const size_t R = std::min<size_t>(MD, rfd_ctx::kPipeRows);
    RFD_HIP(hipMemcpy2DAsync(ps.h_ob, MD * 5 * sizeof(float), ps.ob.p, MD * 5 * sizeof(float), R * 5 * sizeof(float), n, hipMemcpyDeviceToHost, c->d2h_stream));
    RFD_HIP(hipMemcpy2DAsync(ps.h_ol, MD * 10 * sizeof(float), ps.ol.p, MD * 10 * sizeof(float), R * 10 * sizeof(float), n, hipMemcpyDeviceToHost, c->d2h_stream));
    RFD_HIP(hipMemcpyAsync(ps.h_oc, ps.oc.p, n * sizeof(int), hipMemcpyDeviceToHost, c->d2h_stream));
    RFD_HIP(hipMemcpyAsync(ps.h_ot, ps.ot.p, n * sizeof(int), hipMemcpyDeviceToHost, c->d2h_stream));
    RFD_HIP(hipEventRecord(ps.done, c->d2h_stream));
    ps.n = n;
    c->pipe_head = (c->pipe_head + 1) % rfd_ctx::kPipe;
    ++c->pipe_inflight;
    return RFD_OK;
}

int rfd_collect_batch(rfd_ctx *c, rfd_dets *out, int *n_out)
{
    RFD_CHECK_ARG(c && out && out->boxes && out->landmarks && out->count, "null argument");
    if (c->pipe_inflight <= 0) { set_error("no batch in flight"); return RFD_ERR_STATE; }
    rfd_ctx::PipeSlot &ps = c->pipe[c->pipe_tail];
    RFD_HIP(hipEventSynchronize(ps.done));
    if (*c->h_nms_flag) { // drop the batch: its detections cannot be trusted
        c->pipe_tail = (c->pipe_tail + 1) % rfd_ctx::kPipe;
        --c->pipe_inflight;
        return check_nms_flag(c);
    }
    const size_t MD = (size_t)c->cfg.max_det;
    bool more = false;
    for (int i = 0; i < ps.n; ++i)
        if ((size_t)ps.h_oc[i] > (size_t)rfd_ctx::kPipeRows) { // rows beyond the prefix copied by rfd_submit_batch
            const size_t R = rfd_ctx::kPipeRows, k = (size_t)ps.h_oc[i];
            RFD_HIP(hipMemcpyAsync(ps.h_ob + ((size_t)i * MD + R) * 5, (const float *)ps.ob.p + ((size_t)i * MD + R) * 5, (k - R) * 5 * sizeof(float), hipMemcpyDeviceToHost, c->d2h_stream));
            RFD_HIP(hipMemcpyAsync(ps.h_ol + ((size_t)i * MD + R) * 10, (const float *)ps.ol.p + ((size_t)i * MD + R) * 10, (k - R) * 10 * sizeof(float), hipMemcpyDeviceToHost, c->d2h_stream));
            more = true;
        }
    if (more) RFD_HIP(hipStreamSynchronize(c->d2h_stream));
    for (int i = 0; i < ps.n; ++i) {
        const int k = ps.h_oc[i];
        memcpy(out->boxes + (size_t)i * MD * 5, ps.h_ob + (size_t)i * MD * 5, (size_t)k * 5 * sizeof(float));
        memcpy(out->landmarks + (size_t)i * MD * 10, ps.h_ol + (size_t)i * MD * 10, (size_t)k * 10 * sizeof(float));
        out->count[i] = k;
        if (out->total) out->total[i] = ps.h_ot[i];
    }
    if (n_out) *n_out = ps.n;
    c->pipe_tail = (c->pipe_tail + 1) % rfd_ctx::kPipe;
    --c->pipe_inflight;
    return RFD_OK;
}

// ---- stage-level entry points ----
int rfd_preprocess(rfd_ctx *c, const rfd_image *imgs, int n, uint8_t *det_img, float *tensor, float *det_scale)
{
    RFD_CHECK_ARG(c != nullptr, "ctx is null");
    RFD_TRY(check_images(c, imgs, n));
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    std::vector<float> scales;
    RFD_TRY(stage_frames(c, imgs, n, false, scales));
    const size_t npix = (size_t)n * c->cfg.image_h * c->cfg.image_w;
    PreParams pp;
    memset(&pp, 0, sizeof pp);
    pp.imgs = (const PreImage *)c->imgs.p;
    pp.net_h = c->cfg.image_h; pp.net_w = c->cfg.image_w;
    if (det_img) { RFD_TRY(c->scratch[0].reserve(npix * 3)); pp.out_det_img = (uint8_t *)c->scratch[0].p; }
    if (tensor) { RFD_TRY(c->scratch[1].reserve(npix * 3 * sizeof(float))); pp.out_tensor = (float *)c->scratch[1].p; }
    RFD_TRY(launch_preprocess(pp, n, c->stream));
    if (det_img) RFD_HIP(hipMemcpyAsync(det_img, pp.out_det_img, npix * 3, hipMemcpyDeviceToHost, c->stream));
    if (tensor) RFD_HIP(hipMemcpyAsync(tensor, pp.out_tensor, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream));
    if (det_scale) memcpy(det_scale, scales.data(), n * sizeof(float));
    return RFD_OK;
}

int rfd_forward(rfd_ctx *c, const float *tensor, int n, float *const heads[9])
{
    RFD_CHECK_ARG(c && tensor && heads, "null argument");
    for (int i = 0; i < 9; ++i) RFD_CHECK_ARG(heads[i] != nullptr, "head pointer is null");
    if (n < 1 || n > c->cfg.max_batch_size) { set_error("batch %d exceeds max_batch_size %d", n, c->cfg.max_batch_size); return RFD_ERR_CAPACITY; }
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    RFD_TRY(c->ensure_network());
    c->ov_last_n = -1;
    const size_t plane = (size_t)c->cfg.image_h * c->cfg.image_w;
    RFD_TRY(c->scratch[1].reserve(n * plane * 3 * sizeof(float)));
    RFD_HIP(hipMemcpyAsync(c->scratch[1].p, tensor, n * plane * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RFD_TRY(launch_tensor_to_nhwc4((const float *)c->scratch[1].p, (bf16_t *)c->net.tensor_ptr(c->net.g.input), n,
                                   c->cfg.image_h, c->cfg.image_w, c->stream));
    RFD_TRY(c->net.run(n, c->stream));
    RFD_HIP(hipMemcpyAsync(c->h_nms_flag, c->net.d_fail, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    for (int l = 0; l < kNumLevels; ++l) {
        const size_t hw = (size_t)c->fh[l] * c->fw[l];
        RFD_TRY(c->scratch[2].reserve(n * hw * 32 * sizeof(float)));
        float *base = (float *)c->scratch[2].p;
        float *cls = base, *bbox = base + n * hw * 4, *lmk = base + n * hw * 12;
        RFD_TRY(launch_heads_to_nchw((const float *)c->net.tensor_ptr(c->net.g.heads[l]), cls, bbox, lmk, n,
                                     c->fh[l], c->fw[l], c->stream));
        RFD_HIP(hipMemcpyAsync(heads[3 * l + 0], cls, n * hw * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RFD_HIP(hipMemcpyAsync(heads[3 * l + 1], bbox, n * hw * 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RFD_HIP(hipMemcpyAsync(heads[3 * l + 2], lmk, n * hw * 20 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        RFD_HIP(hipStreamSynchronize(c->stream)); // scratch[2] is reused by the next level
    }
    if (c->net.profiling) RFD_TRY(c->net.collect_profile());
    return check_nms_flag(c);
}

int rfd_decode_nms(rfd_ctx *c, const float *const heads[9], int n, const float *det_scale, rfd_dets *out,
                   int32_t *gidx)
{
    RFD_CHECK_ARG(c && heads && det_scale, "null argument");
    RFD_CHECK_ARG(out && out->boxes && out->landmarks && out->count, "output buffers are null");
    for (int i = 0; i < 9; ++i) RFD_CHECK_ARG(heads[i] != nullptr, "head pointer is null");
    if (n < 1 || n > c->cfg.max_batch_size) { set_error("batch %d exceeds max_batch_size %d", n, c->cfg.max_batch_size); return RFD_ERR_CAPACITY; }
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    static const int chans[3] = {2 * kA, 4 * kA, 10 * kA};
    DecodeParams dp;
    fill_decode_params(c, dp);
    for (int l = 0; l < kNumLevels; ++l)
        for (int k = 0; k < 3; ++k) {
            const size_t bytes = (size_t)n * chans[k] * c->fh[l] * c->fw[l] * sizeof(float);
            DevBuf &b = c->scratch[3 + 3 * l + k];
            RFD_TRY(b.reserve(bytes));
            RFD_HIP(hipMemcpyAsync(b.p, heads[3 * l + k], bytes, hipMemcpyHostToDevice, c->stream));
            (k == 0 ? dp.cls[l] : k == 1 ? dp.bbox[l] : dp.lmk[l]) = (const float *)b.p;
        }
    RFD_HIP(hipMemcpyAsync(c->det_scale.p, det_scale, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RFD_HIP(hipEventRecord(c->ev[3], c->stream));
    RFD_TRY(post_network(c, dp, true, n, (float *)c->out_boxes.p, (float *)c->out_lmk.p, (int *)c->out_count.p,
                         (int *)c->out_total.p, gidx ? (int *)c->out_gidx.p : nullptr));
    const size_t MD = (size_t)c->cfg.max_det;
    RFD_HIP(hipMemcpyAsync(out->boxes, c->out_boxes.p, n * MD * 5 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipMemcpyAsync(out->landmarks, c->out_lmk.p, n * MD * 10 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipMemcpyAsync(out->count, c->out_count.p, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (out->total) RFD_HIP(hipMemcpyAsync(out->total, c->out_total.p, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (gidx) RFD_HIP(hipMemcpyAsync(gidx, c->out_gidx.p, n * MD * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipEventRecord(c->ev[7], c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream));
    RFD_TRY(check_nms_flag(c));
    RFD_TRY(finish_stats(c, n, false, false));
    for (int i = 0; i < n; ++i) c->stats.detections += out->total ? out->total[i] : out->count[i];
    return RFD_OK;
}

void rfd_selection_config_default(rfd_selection_config *cfg)
{
    if (!cfg) return;
    cfg->margin_center_left_ratio = 0.3f;  // config.rs:110
    cfg->margin_center_right_ratio = 0.3f; // config.rs:111
    cfg->margin_edge_ratio = 0.1f;         // config.rs:112
    cfg->minimum_face_ratio = 0.0075f;     // config.rs:113
}

// selection over device-resident detection slabs; results copied to the host pointers
static int select_impl(rfd_ctx *c, const float *d_boxes, const float *d_lmk, const int *d_count, const int *img_h,
                       const int *img_w, int n, const rfd_selection_config *cfg, int is_enroll, float *out_box,
                       float *out_kps, int32_t *found)
{
    rfd_selection_config def;
    rfd_selection_config_default(&def);
    if (!cfg) cfg = &def;
    RFD_TRY(c->sel_dims.reserve((size_t)2 * n * sizeof(int)));
    RFD_TRY(c->sel_out.reserve((size_t)n * 16 * sizeof(float)));
    std::vector<int> dims(2 * n);
    for (int i = 0; i < n; ++i) { dims[i] = img_h[i]; dims[n + i] = img_w[i]; }
    RFD_HIP(hipMemcpyAsync(c->sel_dims.p, dims.data(), 2 * n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    SelectParams sp;
    memset(&sp, 0, sizeof sp);
    sp.boxes = d_boxes; sp.lmk = d_lmk; sp.count = d_count;
    sp.img_h = (const int *)c->sel_dims.p; sp.img_w = sp.img_h + n;
    sp.n = n; sp.max_det = c->cfg.max_det; sp.is_enroll = is_enroll;
    sp.margin_center_left_ratio = cfg->margin_center_left_ratio;
    sp.margin_center_right_ratio = cfg->margin_center_right_ratio;
    sp.margin_edge_ratio = cfg->margin_edge_ratio;
    sp.minimum_face_ratio = cfg->minimum_face_ratio;
    float *o = (float *)c->sel_out.p;
    sp.out_box = o; sp.out_kps = o + (size_t)n * 5; sp.out_found = (int *)(o + (size_t)n * 15);
    RFD_TRY(launch_face_select(sp, c->stream));
    RFD_HIP(hipMemcpyAsync(out_box, sp.out_box, (size_t)n * 5 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipMemcpyAsync(out_kps, sp.out_kps, (size_t)n * 10 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipMemcpyAsync(found, sp.out_found, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream)); // dims is a host temporary
    return check_nms_flag(c);
}

int rfd_select_faces(rfd_ctx *c, const rfd_dets *dets, const int *img_h, const int *img_w, int n,
                     const rfd_selection_config *cfg, int is_enroll, float *out_box, float *out_kps, int32_t *found)
{
    RFD_CHECK_ARG(c && dets && dets->boxes && dets->landmarks && dets->count && img_h && img_w && out_box && out_kps && found,
                  "null argument");
    if (n < 1 || n > c->cfg.max_batch_size) { set_error("batch %d exceeds max_batch_size %d", n, c->cfg.max_batch_size); return RFD_ERR_CAPACITY; }
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    const size_t MD = (size_t)c->cfg.max_det;
    RFD_HIP(hipMemcpyAsync(c->out_boxes.p, dets->boxes, n * MD * 5 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RFD_HIP(hipMemcpyAsync(c->out_lmk.p, dets->landmarks, n * MD * 10 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RFD_HIP(hipMemcpyAsync(c->out_count.p, dets->count, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    return select_impl(c, (const float *)c->out_boxes.p, (const float *)c->out_lmk.p, (const int *)c->out_count.p, img_h, img_w,
                       n, cfg, is_enroll, out_box, out_kps, found);
}

int rfd_detect_select_batch(rfd_ctx *c, const rfd_image *imgs, int n, const rfd_selection_config *cfg, int is_enroll,
                            float *out_box, float *out_kps, int32_t *found)
{
    RFD_CHECK_ARG(c && out_box && out_kps && found, "null argument");
    RFD_TRY(check_images(c, imgs, n));
    rfd_dets dev = {(float *)c->out_boxes.p, (float *)c->out_lmk.p, (int32_t *)c->out_count.p, (int32_t *)c->out_total.p};
    RFD_TRY(detect_impl(c, imgs, n, &dev, /*outputs stay on the device*/ true, /*async*/ 1, /*frames on host*/ false));
    std::vector<int> hh(n), ww(n);
    for (int i = 0; i < n; ++i) { hh[i] = imgs[i].height; ww[i] = imgs[i].width; }
    return select_impl(c, dev.boxes, dev.landmarks, dev.count, hh.data(), ww.data(), n, cfg, is_enroll, out_box, out_kps, found);
}

void rfd_alignment_config_default(rfd_alignment_config *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->out_w = 112; cfg->out_h = 112; // config.rs:46
    static const float tmpl[10] = {38.2946f, 51.6963f, 73.5318f, 51.5014f, 56.0252f, 71.7366f, 41.5493f, 92.3655f, 70.7299f, 92.2041f}; // :47-52
    memcpy(cfg->standard_landmarks, tmpl, sizeof tmpl);
}

// alignment of the frames whose descriptors sit in c->imgs (the last staged batch); selection results are device
// arrays; crops and status are copied to the host pointers
static int align_impl(rfd_ctx *c, int n, const float *d_box, const float *d_kps, const int *d_found,
                      const rfd_alignment_config *cfg, uint8_t *out_crops, int32_t *status)
{
    rfd_alignment_config def;
    rfd_alignment_config_default(&def);
    if (!cfg) cfg = &def;
    if (cfg->out_w < 1 || cfg->out_h < 1 || cfg->out_w > 4096 || cfg->out_h > 4096) {
        set_error("alignment output size %dx%d out of range", cfg->out_w, cfg->out_h);
        return RFD_ERR_INVALID_ARG;
    }
    const size_t crop = (size_t)cfg->out_w * cfg->out_h * 3;
    RFD_TRY(c->align_faces.reserve((size_t)n * sizeof(AlignFace)));
    RFD_TRY(c->align_out.reserve((size_t)n * crop));
    RFD_TRY(c->align_status.reserve((size_t)n * sizeof(int)));
    AlignParams ap;
    memset(&ap, 0, sizeof ap);
    ap.imgs = (const PreImage *)c->imgs.p;
    ap.box = d_box; ap.kps = d_kps; ap.found = d_found;
    memcpy(ap.std_lmk, cfg->standard_landmarks, sizeof ap.std_lmk);
    ap.out_w = cfg->out_w; ap.out_h = cfg->out_h; ap.n = n;
    ap.faces = (AlignFace *)c->align_faces.p;
    ap.status = (int *)c->align_status.p;
    ap.out = (uint8_t *)c->align_out.p;
    RFD_TRY(launch_face_align(ap, c->stream));
    RFD_HIP(hipMemcpyAsync(out_crops, ap.out, (size_t)n * crop, hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipMemcpyAsync(status, ap.status, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream));
    return check_nms_flag(c);
}

int rfd_align_faces(rfd_ctx *c, const rfd_image *imgs, int n, const float *boxes, const float *kps, const int32_t *found,
                    const rfd_alignment_config *cfg, uint8_t *out_crops, int32_t *status)
{
    RFD_CHECK_ARG(c && boxes && kps && found && out_crops && status, "null argument");
    RFD_TRY(check_images(c, imgs, n));
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    std::vector<float> scales;
    RFD_TRY(stage_frames(c, imgs, n, false, scales));
    RFD_TRY(c->sel_out.reserve((size_t)n * 16 * sizeof(float)));
    float *o = (float *)c->sel_out.p;
    RFD_HIP(hipMemcpyAsync(o, boxes, (size_t)n * 5 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RFD_HIP(hipMemcpyAsync(o + (size_t)n * 5, kps, (size_t)n * 10 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    RFD_HIP(hipMemcpyAsync(o + (size_t)n * 15, found, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
    return align_impl(c, n, o, o + (size_t)n * 5, (const int *)(o + (size_t)n * 15), cfg, out_crops, status);
}

int rfd_detect_select_align_batch(rfd_ctx *c, const rfd_image *imgs, int n, const rfd_selection_config *sel_cfg, int is_enroll,
                                  const rfd_alignment_config *align_cfg, float *out_box, float *out_kps, int32_t *found,
                                  uint8_t *out_crops, int32_t *status)
{
    RFD_CHECK_ARG(c && out_crops && status, "null argument");
    // the staged frames and their descriptors (c->staging / c->imgs) stay valid until the next call on this context
    RFD_TRY(rfd_detect_select_batch(c, imgs, n, sel_cfg, is_enroll, out_box, out_kps, found));
    const float *o = (const float *)c->sel_out.p;
    return align_impl(c, n, o, o + (size_t)n * 5, (const int *)(o + (size_t)n * 15), align_cfg, out_crops, status);
}

int rfd_nms_sorted(rfd_ctx *c, int32_t *keep, int *num_out, const float *boxes, int boxes_num, int boxes_dim,
                   float thresh)
{
    RFD_CHECK_ARG(c && keep && num_out && (boxes || boxes_num == 0), "null argument");
    RFD_CHECK_ARG(boxes_num >= 0 && boxes_dim >= 4, "boxes_num < 0 or boxes_dim < 4");
    *num_out = 0;
    if (boxes_num == 0) return RFD_OK;
    RFD_HIP(hipSetDevice(c->cfg.device_id));
    std::vector<float4> packed(boxes_num);
    for (int i = 0; i < boxes_num; ++i) {
        const float *b = boxes + (size_t)i * boxes_dim;
        packed[i] = make_float4(b[0], b[1], b[2], b[3]);
    }
    RFD_TRY(c->scratch[0].reserve((size_t)boxes_num * sizeof(float4)));
    RFD_TRY(c->scratch[1].reserve((size_t)boxes_num * sizeof(int) + 2 * sizeof(int)));
    RFD_HIP(hipMemcpyAsync(c->scratch[0].p, packed.data(), boxes_num * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    NmsParams np;
    memset(&np, 0, sizeof np);
    np.sorted_boxes = (const float4 *)c->scratch[0].p;
    np.presorted_n = boxes_num;
    np.total_anchors = boxes_num;
    np.max_det = boxes_num;
    np.iou_thr = thresh;
    int *d_total = (int *)c->scratch[1].p;
    np.out_total = d_total;
    np.out_gidx = d_total + 2;
    RFD_TRY(launch_nms(np, 1, c->stream));
    int total = 0;
    RFD_HIP(hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RFD_HIP(hipStreamSynchronize(c->stream));
    RFD_HIP(hipMemcpy(keep, d_total + 2, (size_t)total * sizeof(int), hipMemcpyDeviceToHost));
    *num_out = total;
    return RFD_OK;
}

void _nms(int32_t *keep, int *num_out, float *boxes, int boxes_num, int boxes_dim, float thresh, int device_id)
{
    if (num_out) *num_out = -1;
    if (device_id < 0 || device_id >= 16 || !keep || !num_out) return;
    std::lock_guard<std::mutex> lk(g_nms_mu);
    if (!g_nms_ctx[device_id]) {
        rfd_config cfg;
        rfd_config_default(&cfg);
        cfg.device_id = device_id;
        cfg.max_det = 1;
        if (rfd_create(&cfg, &g_nms_ctx[device_id]) != RFD_OK) { g_nms_ctx[device_id] = nullptr; return; }
    }
    if (rfd_nms_sorted(g_nms_ctx[device_id], keep, num_out, boxes, boxes_num, boxes_dim, thresh) != RFD_OK) *num_out = -1;
}

} // extern "C"

/*
 * rfd_oracle.c -- CPU restatement (plain C) of the reference's RetinaFace detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (librfd_hip.so, the python host mirror) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned".  The reference (okieraised/rs-face-detection, Rust) cannot be
 * built here (no rustc/cargo, no OpenCV, no Triton) and its tests assert nothing and hold no
 * expected outputs.  This restatement is pinned only by known answers
 * derived by hand from the reference's formulas on the literal inputs of the reference's own
 * print-only tests (tests/golden/kat_*.json).  The OpenCV bilinear resize
 * (third party: opencv crate 0.92.0 -> system OpenCV 4.x, call site face_detection.rs:156) is
 * restated from OpenCV's published algorithm and is unpinned as well.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Arithmetic is IEEE f32 in the written operation order: compile with -ffp-contract=off.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RFD_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Anchors: src/processing/generate_anchors.rs
 * ------------------------------------------------------------------------------------------ */

/* _whctrs, generate_anchors.rs:20-26 */
static void whctrs(const float a[4], float *w, float *h, float *xc, float *yc)
{
    *w = a[2] - a[0] + 1.0f;
    *h = a[3] - a[1] + 1.0f;
    *xc = a[0] + 0.5f * (*w - 1.0f);
    *yc = a[1] + 0.5f * (*h - 1.0f);
}

/* _mkanchors, generate_anchors.rs:28-39 (one row) */
static void mkanchor(float ws, float hs, float xc, float yc, float out[4])
{
    out[0] = xc - 0.5f * (ws - 1.0f);
    out[1] = yc - 0.5f * (hs - 1.0f);
    out[2] = xc + 0.5f * (ws - 1.0f);
    out[3] = yc + 0.5f * (hs - 1.0f);
}

/*
 * generate_anchors2 (dense_anchor = false), generate_anchors.rs:61-93, with _ratio_enum :141-148
 * and _scale_enum :151-157.  out: [n_ratios * n_scales][4], ratio-major.
 */
RFD_API int rfd_oracle_generate_anchors2(int base_size, const float *ratios, int n_ratios,
                                         const float *scales, int n_scales, float *out)
{
    float base[4] = {1.0f - 1.0f, 1.0f - 1.0f, (float)base_size - 1.0f, (float)base_size - 1.0f};
    int n = 0;
    for (int r = 0; r < n_ratios; ++r) {
        float w, h, xc, yc;
        whctrs(base, &w, &h, &xc, &yc);
        float size = w * h;
        float size_ratio = size / ratios[r];
        float ws = roundf(sqrtf(size_ratio)); /* f32::round = half away from zero */
        float hs = ws * ratios[r];
        float ra[4];
        mkanchor(ws, hs, xc, yc, ra);
        float rw, rh, rxc, ryc;
        whctrs(ra, &rw, &rh, &rxc, &ryc);
        for (int s = 0; s < n_scales; ++s) {
            mkanchor(rw * scales[s], rh * scales[s], rxc, ryc, out + 4 * n);
            ++n;
        }
    }
    return n;
}

/*
 * generate_anchors_fpn2 with the production config of face_detection.rs:55-80:
 * strides sorted descending (generate_anchors.rs:123-124) = 32,16,8; base 16; ratio 1;
 * scales {32:(32,16), 16:(8,4), 8:(2,1)}.  out: [3][2][4].
 */
RFD_API void rfd_oracle_anchors_fpn(float *out)
{
    static const float ratio[1] = {1.0f};
    static const float scales[3][2] = {{32.0f, 16.0f}, {8.0f, 4.0f}, {2.0f, 1.0f}};
    for (int l = 0; l < 3; ++l)
        rfd_oracle_generate_anchors2(16, ratio, 1, scales[l], 2, out + l * 8);
}

/* rcnn::anchors::anchors, src/rcnn/anchors.rs:3-21.  out: [height][width][a][4]. */
RFD_API void rfd_oracle_anchor_plane(int height, int width, int stride, const float *base, int a,
                                     float *out)
{
    for (int iw = 0; iw < width; ++iw) {
        float sw = (float)(iw * stride);
        for (int ih = 0; ih < height; ++ih) {
            float sh = (float)(ih * stride);
            for (int k = 0; k < a; ++k) {
                float *o = out + (((size_t)ih * width + iw) * a + k) * 4;
                o[0] = base[k * 4 + 0] + sw;
                o[1] = base[k * 4 + 1] + sh;
                o[2] = base[k * 4 + 2] + sw;
                o[3] = base[k * 4 + 3] + sh;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Box / landmark decode: src/pipeline/module/face_detection.rs:516-570,
 * clip: src/processing/bbox_transform.rs:27-45
 * ------------------------------------------------------------------------------------------ */

/* f32::exp (face_detection.rs:534-535) is the platform libm's expf; every oracle function calls exactly that.  The two entry
 * points below exist to PIN the device's restatement of it (csrc/kernels_post.hip: exp_cr) without a GPU: `restated` is the same
 * operation sequence as the device code (glibc >= 2.27 e_expf.c: exp2f table algorithm, r = fma(x, 32/ln2, -k): the build glibc's
 * ifunc selects on x86-64 CPUs with FMA), `libm` is the host's expf.  tests/test_oracle_cpu.py compares them on ~10^7 inputs. */
static const unsigned long long k_exp2f_tab[32] = {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
RFD_API void rfd_oracle_expf_restated(const float *x, int n, float *out)
{
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    for (int i = 0; i < n; ++i) {
        uint32_t u;
        memcpy(&u, &x[i], 4);
        const uint32_t abstop = (u >> 20) & 0x7ff;
        if (abstop >= 0x42b) { /* |x| >= 88 or NaN */
            if (u == 0xff800000u) { out[i] = 0.0f; continue; }
            if (abstop >= 0x7f8) { out[i] = x[i] + x[i]; continue; }
            if (x[i] > 0x1.62e42ep6f) { out[i] = INFINITY; continue; }
            if (x[i] < -0x1.9fe368p6f) { out[i] = 0.0f; continue; }
        }
        const double xd = (double)x[i];
        double kd = fma(InvLn2N, xd, Shift);
        unsigned long long ki;
        memcpy(&ki, &kd, 8);
        kd -= Shift;
        const double r = fma(InvLn2N, xd, -kd);
        const unsigned long long t = k_exp2f_tab[ki & 31] + (ki << 47);
        double sc;
        memcpy(&sc, &t, 8);
        const double z = fma(C0, r, C1), r2 = r * r;
        double y = fma(C2, r, 1.0);
        y = fma(z, r2, y);
        out[i] = (float)(y * sc);
    }
}
RFD_API void rfd_oracle_expf_libm(const float *x, int n, float *out)
{
    for (int i = 0; i < n; ++i) out[i] = expf(x[i]);
}

/* bbox_pred, face_detection.rs:516-549 (first 4 columns). boxes,deltas,out: [n][4]. */
RFD_API void rfd_oracle_bbox_pred(const float *boxes, const float *deltas, int n, float *out)
{
    for (int i = 0; i < n; ++i) {
        const float *b = boxes + 4 * i, *d = deltas + 4 * i;
        float w = b[2] - b[0] + 1.0f;               /* :522 */
        float h = b[3] - b[1] + 1.0f;               /* :523 */
        float cx = b[0] + 0.5f * (w - 1.0f);        /* :524 */
        float cy = b[1] + 0.5f * (h - 1.0f);        /* :525 */
        float pcx = d[0] * w + cx;                  /* :532 */
        float pcy = d[1] * h + cy;                  /* :533 */
        float pw = expf(d[2]) * w;                  /* :534 (f32::exp -> libm expf) */
        float ph = expf(d[3]) * h;                  /* :535 */
        out[4 * i + 0] = pcx - 0.5f * (pw - 1.0f);  /* :539 */
        out[4 * i + 1] = pcy - 0.5f * (ph - 1.0f);  /* :540 */
        out[4 * i + 2] = pcx + 0.5f * (pw - 1.0f);  /* :541 */
        out[4 * i + 3] = pcy + 0.5f * (ph - 1.0f);  /* :542 */
    }
}

/* landmark_pred, face_detection.rs:551-570. boxes [n][4], deltas/out [n][5][2]. */
RFD_API void rfd_oracle_landmark_pred(const float *boxes, const float *deltas, int n, float *out)
{
    for (int i = 0; i < n; ++i) {
        const float *b = boxes + 4 * i;
        float w = b[2] - b[0] + 1.0f;
        float h = b[3] - b[1] + 1.0f;
        float cx = b[0] + 0.5f * (w - 1.0f);
        float cy = b[1] + 0.5f * (h - 1.0f);
        for (int p = 0; p < 5; ++p) {
            out[i * 10 + 2 * p + 0] = deltas[i * 10 + 2 * p + 0] * w + cx; /* :565 */
            out[i * 10 + 2 * p + 1] = deltas[i * 10 + 2 * p + 1] * h + cy; /* :566 */
        }
    }
}

/* Rust f32::min / f32::max: if one operand is NaN the other is returned. */
static float rmin(float a, float b) { return fminf(a, b); }
static float rmax(float a, float b) { return fmaxf(a, b); }

/* clip_boxes, bbox_transform.rs:27-45: v.min(hi).max(0). boxes [n][4] in place. */
RFD_API void rfd_oracle_clip_boxes(float *boxes, int n, int im_h, int im_w)
{
    float width = (float)im_w - 1.0f, height = (float)im_h - 1.0f;
    for (int i = 0; i < n; ++i) {
        float *b = boxes + 4 * i;
        b[0] = rmax(rmin(b[0], width), 0.0f);
        b[1] = rmax(rmin(b[1], height), 0.0f);
        b[2] = rmax(rmin(b[2], width), 0.0f);
        b[3] = rmax(rmin(b[3], height), 0.0f);
    }
}

/* ------------------------------------------------------------------------------------------
 * Stable argsort descending: src/utils/utils.rs:87-95 (Vec::sort_by is a stable merge sort)
 * ------------------------------------------------------------------------------------------ */
static void merge_sort_desc(const float *key, int *idx, int *tmp, int n)
{
    if (n < 2) return;
    int h = n / 2;
    merge_sort_desc(key, idx, tmp, h);
    merge_sort_desc(key, idx + h, tmp, n - h);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take right only if strictly greater: ties keep the earlier (left) element first */
        if (key[idx[j]] > key[idx[i]]) tmp[k++] = idx[j++];
        else tmp[k++] = idx[i++];
    }
    while (i < h) tmp[k++] = idx[i++];
    while (j < n) tmp[k++] = idx[j++];
    memcpy(idx, tmp, (size_t)n * sizeof(int));
}

RFD_API void rfd_oracle_argsort_desc(const float *scores, int n, int *order)
{
    int *tmp = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    for (int i = 0; i < n; ++i) order[i] = i;
    merge_sort_desc(scores, order, tmp, n);
    free(tmp);
}

/* ------------------------------------------------------------------------------------------
 * Greedy NMS: src/processing/nms.rs:3-65.  dets [n][5] = x1,y1,x2,y2,score.
 * Returns the number kept; keep[] = indices into dets in kept order.
 * The survivor rule is `ovr <= thresh` (:58): a NaN overlap is suppressed.
 * ------------------------------------------------------------------------------------------ */
RFD_API int rfd_oracle_nms(const float *dets, int n, float thresh, int *keep)
{
    if (n <= 0) return 0;
    float *sc = (float *)malloc((size_t)n * sizeof(float));
    int *order = (int *)malloc((size_t)n * sizeof(int));
    for (int i = 0; i < n; ++i) sc[i] = dets[5 * i + 4];
    rfd_oracle_argsort_desc(sc, n, order); /* :5-6 (stable; a no-op on pre-sorted input) */
    int m = n, nk = 0;
    while (m > 0) {
        int i = order[0];
        keep[nk++] = i; /* :11-12 */
        const float *bi = dets + 5 * i;
        float area_i = (bi[2] - bi[0] + 1.0f) * (bi[3] - bi[1] + 1.0f); /* :46 */
        int m2 = 0;
        for (int t = 1; t < m; ++t) {
            int j = order[t];
            const float *bj = dets + 5 * j;
            float xx1 = rmax(bi[0], bj[0]); /* :14-19 */
            float yy1 = rmax(bi[1], bj[1]);
            float xx2 = rmin(bi[2], bj[2]);
            float yy2 = rmin(bi[3], bj[3]);
            float w = xx2 - xx1 + 1.0f;     /* :39 */
            float h = yy2 - yy1 + 1.0f;     /* :40 */
            w = rmax(0.0f, w);              /* :42 */
            h = rmax(0.0f, h);              /* :43 */
            float inter = w * h;            /* :45 */
            float area_j = (bj[2] - bj[0] + 1.0f) * (bj[3] - bj[1] + 1.0f); /* :50 */
            float ovr = inter / (area_i + area_j - inter);                  /* :54 */
            if (ovr <= thresh) order[m2++] = j;                             /* :58-61 */
        }
        m = m2;
    }
    free(sc);
    free(order);
    return nk;
}

/* ------------------------------------------------------------------------------------------
 * _forward after the network: face_detection.rs:319-470, and _postprocess :473-493.
 *
 * heads[9]: per level (stride 32,16,8) cls [2A,h,w], bbox [4A,h,w], lmk [10A,h,w], f32, NCHW with
 * N = 1 (the reference's Triton output contract, face_detection.rs:286-312), A = 2.
 * net_h, net_w: the network input size (im_info, :213-218; clip bound :373).
 * det_scale <= 0 means "do not rescale" (skip _postprocess).
 * Outputs: det [K][5], lmk [K][5][2], gidx [K] = global anchor row index of each kept detection
 * (level offset + (h*W+w)*A+a).  cap = capacity in rows of the output buffers.
 * Returns K (the true count, which may exceed cap; only min(K,cap) rows are written), or -1.
 * n_candidates (optional) receives the number of rows with score >= conf_thr.
 * ------------------------------------------------------------------------------------------ */
RFD_API int rfd_oracle_decode_nms(const float *const *heads, int net_h, int net_w, float conf_thr,
                                  float iou_thr, float det_scale, float *det, float *lmk,
                                  int *gidx, int cap, int *n_candidates)
{
    static const int strides[3] = {32, 16, 8};
    const int A = 2;
    float base[3 * 2 * 4];
    rfd_oracle_anchors_fpn(base);

    int total = 0;
    for (int l = 0; l < 3; ++l) total += (net_h / strides[l]) * (net_w / strides[l]) * A;
    float *props = (float *)malloc((size_t)total * 4 * sizeof(float));
    float *scores = (float *)malloc((size_t)total * sizeof(float));
    float *lmks = (float *)malloc((size_t)total * 10 * sizeof(float));
    int *gids = (int *)malloc((size_t)total * sizeof(int));
    if (!props || !scores || !lmks || !gids) return -1;

    int n = 0, goff = 0;
    for (int l = 0; l < 3; ++l) { /* :319 */
        const int s = strides[l];
        const int fh = net_h / s, fw = net_w / s, k = fh * fw;
        const float *cls = heads[3 * l + 0], *bbx = heads[3 * l + 1], *lmd = heads[3 * l + 2];
        float *plane = (float *)malloc((size_t)k * A * 4 * sizeof(float));
        float *deltas = (float *)malloc((size_t)k * A * 4 * sizeof(float));
        float *boxes = (float *)malloc((size_t)k * A * 4 * sizeof(float));
        float *ldel = (float *)malloc((size_t)k * A * 10 * sizeof(float));
        float *lpred = (float *)malloc((size_t)k * A * 10 * sizeof(float));
        rfd_oracle_anchor_plane(fh, fw, s, base + l * 8, A, plane); /* :329 */
        for (int h = 0; h < fh; ++h)
            for (int w = 0; w < fw; ++w)
                for (int a = 0; a < A; ++a) {
                    int r = (h * fw + w) * A + a; /* NHWC flatten, :336-364 */
                    for (int c = 0; c < 4; ++c)
                        deltas[r * 4 + c] = bbx[((size_t)(4 * a + c) * fh + h) * fw + w] * 1.0f; /* bbox_stds :366-371 */
                    for (int c = 0; c < 10; ++c)
                        ldel[r * 10 + c] = lmd[((size_t)(10 * a + c) * fh + h) * fw + w] * 1.0f; /* landmark_std :398 */
                }
        rfd_oracle_bbox_pred(plane, deltas, k * A, boxes);      /* :372 */
        rfd_oracle_clip_boxes(boxes, k * A, net_h, net_w);      /* :373 */
        rfd_oracle_landmark_pred(plane, ldel, k * A, lpred);    /* :399 */
        for (int h = 0; h < fh; ++h)
            for (int w = 0; w < fw; ++w)
                for (int a = 0; a < A; ++a) {
                    int r = (h * fw + w) * A + a;
                    float sc = cls[((size_t)(A + a) * fh + h) * fw + w]; /* :322 fg = channel A+a */
                    if (sc >= conf_thr) {                                /* :375 */
                        memcpy(props + 4 * n, boxes + 4 * r, 4 * sizeof(float));
                        memcpy(lmks + 10 * n, lpred + 10 * r, 10 * sizeof(float));
                        scores[n] = sc;
                        gids[n] = goff + r;
                        ++n;
                    }
                }
        goff += k * A;
        free(plane); free(deltas); free(boxes); free(ldel); free(lpred);
    }
    if (n_candidates) *n_candidates = n;

    int K = 0;
    if (n > 0) { /* :413-419 empty early-out otherwise */
        int *order = (int *)malloc((size_t)n * sizeof(int));
        rfd_oracle_argsort_desc(scores, n, order); /* :423 */
        float *pre = (float *)malloc((size_t)n * 5 * sizeof(float));
        for (int i = 0; i < n; ++i) { /* :424-430 */
            memcpy(pre + 5 * i, props + 4 * order[i], 4 * sizeof(float));
            pre[5 * i + 4] = scores[order[i]];
        }
        int *keep = (int *)malloc((size_t)n * sizeof(int));
        K = rfd_oracle_nms(pre, n, iou_thr, keep); /* :431 */
        for (int t = 0; t < K && t < cap; ++t) {   /* :433-464 */
            int i = keep[t], src = order[i];
            for (int c = 0; c < 4; ++c)
                det[5 * t + c] = det_scale > 0.0f ? pre[5 * i + c] / det_scale : pre[5 * i + c]; /* :477-481 */
            det[5 * t + 4] = pre[5 * i + 4];
            for (int c = 0; c < 10; ++c)
                lmk[10 * t + c] = det_scale > 0.0f ? lmks[10 * src + c] / det_scale : lmks[10 * src + c]; /* :483 */
            if (gidx) gidx[t] = gids[src];
        }
        free(order); free(pre); free(keep);
    }
    free(props); free(scores); free(lmks); free(gids);
    return K;
}

/* ------------------------------------------------------------------------------------------
 * _preprocess: face_detection.rs:131-198 (geometry) + OpenCV resize INTER_LINEAR 8UC3 (:156)
 * ------------------------------------------------------------------------------------------ */

/* face_detection.rs:140-153. image_size = (w, h). */
RFD_API void rfd_oracle_geometry(int img_h, int img_w, int size_w, int size_h, int *new_w,
                                 int *new_h, float *det_scale)
{
    float im_ratio = (float)img_h / (float)img_w;
    float model_ratio = (float)size_h / (float)size_w;
    if (im_ratio > model_ratio) {
        *new_h = size_h;
        *new_w = (int)((float)*new_h / im_ratio);
    } else {
        *new_w = size_w;
        *new_h = (int)((float)*new_w * im_ratio);
    }
    *det_scale = (float)*new_h / (float)img_h;
}

static short sat_short_round(float v)
{
    /* cv::saturate_cast<short>(float) = saturate(cvRound(v)); cvRound rounds half to even */
    long r = lrintf(v);
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return (short)r;
}

/*
 * cv::resize(src, dst, Size(dw,dh), 0, 0, INTER_LINEAR) for CV_8UC3 (third-party; restated from
 * OpenCV 4.x imgproc/resize.cpp; unpinned).  Coefficients: fx = (float)((dx+0.5)*scale_x-0.5),
 * scale_x = 1/((double)dw/sw); 11-bit fixed point (INTER_RESIZE_COEF_SCALE = 2048);
 * horizontal pass in int, vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
 * When both scale factors are exactly 2 OpenCV switches INTER_LINEAR to INTER_AREA whose fast
 * 2x2 path is (a+b+c+d+2)>>2.
 */
RFD_API void rfd_oracle_resize_linear_u8c3(const uint8_t *src, int sh, int sw, ptrdiff_t sstride,
                                           uint8_t *dst, int dh, int dw, ptrdiff_t dstride)
{
    const int cn = 3;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
    int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y); /* saturate_cast<int> */
    int is_area_fast = fabs(scale_x - iscale_x) < 2.220446049250313e-16 &&
                       fabs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
        for (int dy = 0; dy < dh; ++dy) {
            const uint8_t *s0 = src + (ptrdiff_t)(2 * dy) * sstride;
            const uint8_t *s1 = s0 + sstride;
            uint8_t *d = dst + (ptrdiff_t)dy * dstride;
            for (int dx = 0; dx < dw; ++dx)
                for (int c = 0; c < cn; ++c) {
                    int i = 2 * dx * cn + c;
                    d[dx * cn + c] = (uint8_t)((s0[i] + s0[i + cn] + s1[i] + s1[i + cn] + 2) >> 2);
                }
        }
        return;
    }
    int *xofs = (int *)malloc((size_t)dw * sizeof(int));
    short *ialpha = (short *)malloc((size_t)dw * 2 * sizeof(short));
    int *rows0 = (int *)malloc((size_t)dw * cn * sizeof(int));
    int *rows1 = (int *)malloc((size_t)dw * cn * sizeof(int));
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.0f; sx = 0; }
        if (sx >= sw - 1) { fx = 0.0f; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx + 0] = sat_short_round((1.0f - fx) * 2048.0f);
        ialpha[2 * dx + 1] = sat_short_round(fx * 2048.0f);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        short b0 = sat_short_round((1.0f - fy) * 2048.0f);
        short b1 = sat_short_round(fy * 2048.0f);
        int sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const uint8_t *S0 = src + (ptrdiff_t)sy0 * sstride, *S1 = src + (ptrdiff_t)sy1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1; /* a1 == 0 whenever sx is clamped */
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            for (int c = 0; c < cn; ++c) {
                rows0[dx * cn + c] = S0[sx * cn + c] * a0 + S0[sx1 * cn + c] * a1;
                rows1[dx * cn + c] = S1[sx * cn + c] * a0 + S1[sx1 * cn + c] * a1;
            }
        }
        uint8_t *d = dst + (ptrdiff_t)dy * dstride;
        for (int x = 0; x < dw * cn; ++x)
            d[x] = (uint8_t)((((b0 * (rows0[x] >> 4)) >> 16) + ((b1 * (rows1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(rows0); free(rows1);
}

/*
 * _preprocess (face_detection.rs:131-198) + tensorise loop (:220-232).
 * src: HxWx3 u8 BGR with row stride sstride.  det_img (optional): size_h x size_w x 3 u8.
 * tensor (optional): [3][size_h][size_w] f32, channel order R,G,B, raw 0..255 values
 * (pixel_scale 1, mean 0, std 1, :105-107, :227).  Returns det_scale.
 */
RFD_API float rfd_oracle_preprocess(const uint8_t *src, int img_h, int img_w, ptrdiff_t sstride,
                                    int size_w, int size_h, uint8_t *det_img, float *tensor)
{
    int new_w, new_h;
    float det_scale;
    rfd_oracle_geometry(img_h, img_w, size_w, size_h, &new_w, &new_h, &det_scale);
    uint8_t *canvas = det_img ? det_img : (uint8_t *)malloc((size_t)size_h * size_w * 3);
    memset(canvas, 0, (size_t)size_h * size_w * 3); /* :169 zero canvas */
    if (new_w > 0 && new_h > 0)
        rfd_oracle_resize_linear_u8c3(src, img_h, img_w, sstride, canvas, new_h, new_w,
                                      (ptrdiff_t)size_w * 3); /* :156 + paste at (0,0) :176-183 */
    if (tensor) {
        for (int i = 0; i < 3; ++i) /* :223-230 */
            for (int y = 0; y < size_h; ++y)
                for (int x = 0; x < size_w; ++x) {
                    uint8_t p = canvas[((size_t)y * size_w + x) * 3 + (2 - i)];
                    tensor[((size_t)i * size_h + y) * size_w + x] = ((float)p / 1.0f - 0.0f) / 1.0f;
                }
    }
    if (!det_img) free(canvas);
    return det_scale;
}

/* ------------------------------------------------------------------------------------------
 * FaceSelection::call, src/pipeline/module/face_selection.rs:72-189, with
 * get_biggest_area_face :28-53.  boxes [k][5], kps [k][5][2] (the detector's outputs), image size of
 * the SOURCE frame.  Returns 1 and fills out_box[5] / out_kps[10] (kps_found = 0 when no key points
 * were matched, :153-176) or 0 when nothing was selected.  The reference's quirks are kept on purpose:
 * "area" is (x_max - x_min)^2 (:113), strict `>` keeps the FIRST maximum, and the key points are those of
 * the first detection within 2 px of the chosen box (:163-173), not necessarily the chosen row.
 * ------------------------------------------------------------------------------------------ */
RFD_API int rfd_oracle_face_selection(const float *boxes, const float *kps, int k, int img_h, int img_w,
                                      float margin_center_left_ratio, float margin_center_right_ratio,
                                      float margin_edge_ratio, float minimum_face_ratio, int is_enroll,
                                      float *out_box, float *out_kps, int *kps_found)
{
    *kps_found = 0;
    if (is_enroll) { /* :84-105: biggest (xmax-xmin)*(ymax-ymin), strict >, from 0.0 */
        float biggest = 0.0f;
        int sel = -1;
        for (int i = 0; i < k; ++i) {
            const float *b = boxes + 5 * i;
            if ((b[2] - b[0]) * (b[3] - b[1]) > biggest) {
                biggest = (b[2] - b[0]) * (b[3] - b[1]);
                sel = i;
            }
        }
        if (sel < 0) return 0;
        memcpy(out_box, boxes + 5 * sel, 5 * sizeof(float));
        memcpy(out_kps, kps + 10 * sel, 10 * sizeof(float));
        *kps_found = 1;
        return 1;
    }
    const float W = (float)img_w, H = (float)img_h;
    const float margin_center_left = margin_center_left_ratio * W;   /* :107 */
    const float margin_center_right = margin_center_right_ratio * W; /* :108 */
    float margin_edge = margin_edge_ratio * W;                        /* :109 */
    margin_edge = fminf(50.0f, margin_edge);                          /* :110 */
    const float x_cen = W / 2.0f;                                     /* :112 */
    unsigned char *valid = (unsigned char *)calloc((size_t)(k > 0 ? k : 1), 1);
    unsigned char *center = (unsigned char *)calloc((size_t)(k > 0 ? k : 1), 1);
    int n_valid = 0, n_center = 0;
    for (int i = 0; i < k; ++i) { /* :114-131 */
        const float *d = boxes + 5 * i;
        const float area = (d[2] - d[0]) * (d[2] - d[0]);
        const float bcw = (d[0] + d[2]) / 2.0f, bch = (d[1] + d[3]) / 2.0f;
        if (bcw >= margin_edge && bcw <= W - margin_edge && bch >= margin_edge && bch <= H - margin_edge &&
            area / (H * W) >= minimum_face_ratio) {
            valid[i] = 1;
            ++n_valid;
        }
    }
    for (int i = 0; i < k; ++i) { /* :133-139 */
        if (!valid[i]) continue;
        const float bcw = (boxes[5 * i] + boxes[5 * i + 2]) / 2.0f;
        if (-margin_center_left <= bcw - x_cen && bcw - x_cen <= margin_center_right) {
            center[i] = 1;
            ++n_center;
        }
    }
    const unsigned char *pool = center; /* :141-147 */
    if (n_center == 0) pool = n_valid == 0 ? NULL : valid;
    float max_size = 0.0f;
    int sel = -1;
    for (int i = 0; i < k; ++i) { /* :152-158 */
        if (pool && !pool[i]) continue;
        const float *r = boxes + 5 * i;
        const float tem = (r[2] - r[0]) + (r[3] - r[1]);
        if (tem > max_size) {
            max_size = tem;
            sel = i;
        }
    }
    free(valid);
    free(center);
    if (sel < 0) return 0; /* :159-161 */
    memcpy(out_box, boxes + 5 * sel, 5 * sizeof(float));
    for (int i = 0; i < k; ++i) { /* :163-180 */
        const float *b = boxes + 5 * i;
        if (fabsf(out_box[0] - b[0]) <= 2.0f && fabsf(out_box[1] - b[1]) <= 2.0f && fabsf(out_box[2] - b[2]) <= 2.0f &&
            fabsf(out_box[3] - b[3]) <= 2.0f) {
            memcpy(out_kps, kps + 10 * i, 10 * sizeof(float));
            *kps_found = 1;
            break;
        }
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * FaceAlignment::call -- src/pipeline/module/face_alignment.rs:27-141, with the 112x112
 * template of FaceAlignmentConfig::new (src/pipeline/face_pipeline/config.rs:46-52).
 *
 * Third-party arithmetic, "parity unpinned" (no OpenCV in this image, the reference's test is commented out,
 * face_alignment.rs:148-240):
 *  - estimate_affine_partial_2d(landmarks -> template, LMEDS, 3.0, 2000, 0.99, 10) (:48-60) is OpenCV calib3d
 *    (modules/calib3d/src/ptsetreg.cpp, OpenCV 4.x; the crate pins opencv 0.92 -> system OpenCV 4, unpinned).
 *    Round 4 restates its published algorithm instead of replacing it by an all-points least squares (round-3 review:
 *    with one bad landmark LMedS drops it and the closed form does not -- and even on clean points the inlier rule
 *    below often keeps 3 or 4 of the 5):
 *      LMeDSPointSetRegistrator::run -- niters = RANSACUpdateNumIters(0.99, outlier ratio 0.45, 2 model points, 2000)
 *        = 13; cv::RNG seeded with (uint64)-1 on EVERY call, so the 13 two-point samples are the same index pairs for
 *        every face (getSubset: uniform(0, count) draws, a repeated index is redrawn; two points are never "collinear");
 *        per sample the exact similarity through the two pairs (AffinePartial2DEstimatorCallback::runKernel, f64),
 *        squared reprojection errors of all points in f32 (computeError), their median; the sample with the smallest
 *        median wins (strict <, so NaN models of coincident points never do); sigma = 2.5 * 1.4826 * (1 + 5 / (count - 2))
 *        * sqrt(min median), at least 0.001; inliers: error <= (float)(sigma^2); fewer than 2 inliers -> empty matrix.
 *      refinement (count > 2, refineIters = 10): cv::LMSolver on the inliers minimises the squared reprojection error of
 *        [a -b tx; b a ty] -- a LINEAR least-squares problem: Levenberg-Marquardt drops its damping after the first
 *        accepted step (lambda 1 -> 0.5 < 0.75 -> 0) and the next, undamped Gauss-Newton step lands on the unique minimum,
 *        the closed form below over the inliers.  DOCUMENTED DIVERGENCE (~1e-12): the closed form stands in for those
 *        iterations.
 *    Written from the published sources as known, not checked against a running OpenCV: parity unpinned.
 *  - warp_affine(INTER_LINEAR, BORDER_CONSTANT, 0) (:112-120) restates cv::warpAffine (imgwarp.cpp): the 2x3
 *    matrix is inverted in f64, coordinates are 10-bit fixed point (AB_BITS) reduced to 5 fractional bits
 *    (INTER_BITS), bilinear weights are the 15-bit table products, out-of-image taps read 0, result
 *    (sum + 2^14) >> 15.
 * ------------------------------------------------------------------------------------------ */
/* closed-form least-squares similarity over the points with use[i] != 0 (use == NULL: all) */
static int similarity_ls(const float *src, const float *dst, int n, const unsigned char *use, double M[6])
{
    double msx = 0, msy = 0, mdx = 0, mdy = 0;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (use && !use[i]) continue;
        msx += (double)src[2 * i]; msy += (double)src[2 * i + 1];
        mdx += (double)dst[2 * i]; mdy += (double)dst[2 * i + 1];
        ++m;
    }
    if (m == 0) return 0;
    msx /= m; msy /= m; mdx /= m; mdy /= m;
    double sxx = 0, sa = 0, sb = 0;
    for (int i = 0; i < n; ++i) {
        if (use && !use[i]) continue;
        const double xs = (double)src[2 * i] - msx, ys = (double)src[2 * i + 1] - msy;
        const double xd = (double)dst[2 * i] - mdx, yd = (double)dst[2 * i + 1] - mdy;
        sxx += xs * xs + ys * ys;
        sa += xs * xd + ys * yd;
        sb += xs * yd - ys * xd;
    }
    if (!(sxx > 0.0)) return 0;
    const double a = sa / sxx, b = sb / sxx;
    M[0] = a; M[1] = -b; M[2] = mdx - (a * msx - b * msy);
    M[3] = b; M[4] = a;  M[5] = mdy - (b * msx + a * msy);
    return 1;
}

/* cv::RNG (core/operations.hpp): multiply-with-carry, CV_RNG_COEFF 4164903690 */
static unsigned cv_rng_next(uint64_t *state)
{
    *state = (uint64_t)(unsigned)*state * 4164903690u + (unsigned)(*state >> 32);
    return (unsigned)*state;
}
static int cv_rng_uniform(uint64_t *state, int a, int b) { return a == b ? a : (int)(cv_rng_next(state) % (unsigned)(b - a)) + a; }

/* RANSACUpdateNumIters (ptsetreg.cpp) */
RFD_API int rfd_oracle_cv_ransac_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p < 0 ? 0 : (p > 1 ? 1 : p);
    ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* AffinePartial2DEstimatorCallback::runKernel: the similarity through two point pairs */
static void similarity_2pt(const float *src, const float *dst, int i0, int i1, double M[6])
{
    const double x1 = src[2 * i0], y1 = src[2 * i0 + 1], x2 = src[2 * i1], y2 = src[2 * i1 + 1];
    const double X1 = dst[2 * i0], Y1 = dst[2 * i0 + 1], X2 = dst[2 * i1], Y2 = dst[2 * i1 + 1];
    const double d = 1. / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
    const double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
    const double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
    const double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
    const double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
    M[0] = S0; M[1] = -S1; M[2] = S2; M[3] = S1; M[4] = S0; M[5] = S3;
}

/* AffinePartial2DEstimatorCallback::computeError: squared reprojection error, f32 */
static void similarity_errors(const float *src, const float *dst, int n, const double M[6], float *err)
{
    const float F0 = (float)M[0], F1 = (float)M[1], F2 = (float)M[2], F3 = (float)M[3], F4 = (float)M[4], F5 = (float)M[5];
    for (int i = 0; i < n; ++i) {
        const float fx = src[2 * i], fy = src[2 * i + 1];
        const float a = F0 * fx + F1 * fy + F2 - dst[2 * i];
        const float b = F3 * fx + F4 * fy + F5 - dst[2 * i + 1];
        err[i] = a * a + b * b;
    }
}

#define RFD_ORACLE_MAX_PTS 64
/* the index pairs LMeDS samples for `n` points (at most `cap` pairs written); returns their number */
RFD_API int rfd_oracle_lmeds_samples(int n, int *pairs, int cap)
{
    if (n <= 2) return 0;
    int niters = rfd_oracle_cv_ransac_num_iters(0.99, 0.45, 2, 2000);
    if (niters < 3) niters = 3;
    uint64_t rng = 0xffffffffffffffffull; /* RNG rng((uint64)-1) */
    int k = 0;
    for (int it = 0; it < niters && k < cap; ++it) {
        const int i0 = cv_rng_uniform(&rng, 0, n);
        int i1;
        do i1 = cv_rng_uniform(&rng, 0, n); while (i1 == i0);
        pairs[2 * k] = i0; pairs[2 * k + 1] = i1;
        ++k;
    }
    return k;
}

/* cv::estimateAffinePartial2D(src -> dst, LMEDS, 3.0, 2000, 0.99, 10) restated (see the block comment above).
 * returns 1 and the 2x3 matrix, or 0: the reference's `transformation_matrix.empty()` branch.  inl (optional): the inlier mask */
RFD_API int rfd_oracle_estimate_similarity_lmeds(const float *src, const float *dst, int n, double M[6], unsigned char *inl)
{
    unsigned char mask[RFD_ORACLE_MAX_PTS];
    float err[RFD_ORACLE_MAX_PTS], srt[RFD_ORACLE_MAX_PTS];
    if (n < 2 || n > RFD_ORACLE_MAX_PTS) return 0;
    if (n == 2) { /* count == modelPoints: the kernel alone, every point an inlier, no refinement */
        similarity_2pt(src, dst, 0, 1, M);
        if (inl) inl[0] = inl[1] = 1;
        return 1;
    }
    int pairs[2 * 64];
    const int ns = rfd_oracle_lmeds_samples(n, pairs, 64);
    double best[6] = {0, 0, 0, 0, 0, 0}, min_median = DBL_MAX;
    for (int s = 0; s < ns; ++s) {
        double Ms[6];
        similarity_2pt(src, dst, pairs[2 * s], pairs[2 * s + 1], Ms);
        similarity_errors(src, dst, n, Ms, err);
        /* std::nth_element(errf.ptr<int>(), ... + count / 2, ...): the f32 errors ordered by their BIT PATTERNS as ints (they are
         * >= +0, or NaN -- which then ranks above every number); the median is element count / 2 of that order */
        for (int i = 0; i < n; ++i) srt[i] = err[i];
        for (int i = 1; i < n; ++i) {
            const float v = srt[i];
            int32_t vb, jb;
            memcpy(&vb, &v, 4);
            int j = i - 1;
            while (j >= 0 && (memcpy(&jb, &srt[j], 4), jb > vb)) { srt[j + 1] = srt[j]; --j; }
            srt[j + 1] = v;
        }
        const double median = (double)srt[n / 2];
        if (median < min_median) { min_median = median; memcpy(best, Ms, sizeof best); }
    }
    if (!(min_median < DBL_MAX)) return 0;
    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 2)) * sqrt(min_median);
    if (!(sigma > 0.001)) sigma = 0.001;
    const float t = (float)(sigma * sigma);
    similarity_errors(src, dst, n, best, err);
    int cnt = 0;
    for (int i = 0; i < n; ++i) { mask[i] = err[i] <= t; cnt += mask[i]; }
    if (inl) memcpy(inl, mask, (size_t)n);
    if (cnt < 2) return 0;
    memcpy(M, best, sizeof best);
    double R[6];
    if (similarity_ls(src, dst, n, mask, R)) memcpy(M, R, sizeof R); /* the fixed point of the LM refinement on the inliers */
    return 1;
}

RFD_API int rfd_oracle_estimate_similarity(const float *src, const float *dst, int n, double M[6])
{
    return rfd_oracle_estimate_similarity_lmeds(src, dst, n, M, NULL);
}

/* the all-points closed form (rounds 1-3; what LMedS + refinement gives when every point is an inlier) */
RFD_API int rfd_oracle_estimate_similarity_all_points(const float *src, const float *dst, int n, double M[6])
{
    return similarity_ls(src, dst, n, NULL, M);
}

static int cv_round_sat(double v) /* cv::saturate_cast<int>(double) = cvRound: nearest, ties to even */
{
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return (int)lrint(v);
}

RFD_API void rfd_oracle_warp_affine_u8c3(const unsigned char *src, int h, int w, ptrdiff_t stride, const double Mfwd[6],
                                         unsigned char *dst, int dh, int dw)
{
    double M[6];
    memcpy(M, Mfwd, sizeof M);
    { /* invert (no WARP_INVERSE_MAP) */
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    const int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, TAB = 1 << INTER_BITS;
    const int round_delta = AB_SCALE / TAB / 2;
    for (int y = 0; y < dh; ++y) {
        const int X0 = cv_round_sat((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        const int Y0 = cv_round_sat((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < dw; ++x) {
            const int adelta = cv_round_sat(M[0] * x * AB_SCALE), bdelta = cv_round_sat(M[3] * x * AB_SCALE);
            const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
            int sx = X >> INTER_BITS, sy = Y >> INTER_BITS; /* saturate_cast<short> */
            sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
            sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
            const int fx = X & (TAB - 1), fy = Y & (TAB - 1);
            int wt[4] = {(TAB - fy) * (TAB - fx) * 32, (TAB - fy) * fx * 32, fy * (TAB - fx) * 32, fy * fx * 32};
            if (fx == 0 && fy == 0) { wt[0] = 32767; wt[3] = 1; } /* saturate_cast<short>(32768) + the table's sum fix-up */
            unsigned char *d = dst + ((size_t)y * dw + x) * 3;
            for (int c = 0; c < 3; ++c) {
                int acc = 0;
                for (int t = 0; t < 4; ++t) {
                    const int px = sx + (t & 1), py = sy + (t >> 1);
                    const int v = (px >= 0 && px < w && py >= 0 && py < h) ? src[(ptrdiff_t)py * stride + px * 3 + c] : 0;
                    acc += v * wt[t];
                }
                d[c] = (unsigned char)((acc + (1 << 14)) >> 15);
            }
        }
    }
}

/* returns 0: aligned by the similarity warp; 1: the crop + resize fallback (:62-110) was taken; -1: the fallback's
 * Rect leaves the image (Mat::roi error, :91-94).  box = selected detection or NULL (:64-71). */
RFD_API int rfd_oracle_face_alignment(const unsigned char *src, int h, int w, ptrdiff_t stride, const float *box,
                                      const float *kps, const float *std_lmk, int out_w, int out_h, unsigned char *dst)
{
    double M[6];
    if (kps && rfd_oracle_estimate_similarity(kps, std_lmk, 5, M)) {
        rfd_oracle_warp_affine_u8c3(src, h, w, stride, M, dst, out_h, out_w);
        return 0;
    }
    float det[4];
    if (!box) { /* :65-69 */
        det[0] = (float)w * 0.0625f; det[1] = (float)h * 0.0625f;
        det[2] = (float)w - det[0]; det[3] = (float)h - det[1];
    } else {
        memcpy(det, box, sizeof det);
    }
    const float margin = 44.0f;
    const float bb0 = fmaxf(det[0] - margin / 2.0f, 0.0f), bb1 = fmaxf(det[1] - margin / 2.0f, 0.0f);
    const float bb2 = fmaxf(det[2] + margin / 2.0f, (float)w); /* `max`, as written (:77) */
    const float bb3 = fmaxf(det[1] + margin / 2.0f, (float)h); /* det[1], as written (:78) */
    const int x0 = (int)bb0, y0 = (int)bb1, x1 = (int)bb2, y1 = (int)bb3;
    const int rw = x1 - x0, rh = y1 - y0;
    if (rw <= 0 || rh <= 0 || x0 + rw > w || y0 + rh > h) return -1;
    rfd_oracle_resize_linear_u8c3(src + (ptrdiff_t)y0 * stride + x0 * 3, rh, rw, stride, dst, out_h, out_w, (ptrdiff_t)out_w * 3);
    return 1;
}

// kernels_pre.hip -- letterbox resize + tensorise on the GPU (gfx950).
//
// Replaces RetinaFaceDetection::_preprocess (reference face_detection.rs:131-198: cv::resize
// INTER_LINEAR on CV_8UC3, paste at the top-left of a zero canvas) and the tensorise loop
// (face_detection.rs:220-232: HWC BGR u8 -> planar RGB f32, raw 0..255).  One pass: each thread
// produces one canvas pixel and writes it in up to three layouts.  HBM-bound byte work:
// reads ~4 source pixels, writes 8 B (NHWC4 bf16) per pixel.
//
// The bilinear arithmetic restates OpenCV's 8-bit fixed-point path (resize.cpp: 11-bit
// coefficients, int horizontal pass, (((b0*(S0>>4))>>16)+((b1*(S1>>4))>>16)+2)>>2 vertical pass,
// and the INTER_AREA 2x2 mean when both scale factors are exactly 2).  Integer-exact.
#include "kernels.h"

namespace rfd {

__device__ __forceinline__ int coef11(float v)
{
    // cv::saturate_cast<short>(v): round half to even, saturate
    int r = __float2int_rn(v);
    return max(-32768, min(32767, r));
}

// one destination pixel of cv::resize(INTER_LINEAR) on CV_8UC3 (see the header): v = B,G,R
__device__ __forceinline__ void resize_linear_px(const uint8_t *src, long long stride, int h, int w, double scale_x,
                                                 double scale_y, int area_fast, int x, int y, int v[3])
{
    if (area_fast) {
        const uint8_t *s0 = src + (long long)(2 * y) * stride + 2 * x * 3;
        const uint8_t *s1 = s0 + stride;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (s0[c] + s0[c + 3] + s1[c] + s1[c + 3] + 2) >> 2;
        return;
    }
    float fx = (float)(((double)x + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= w - 1) { fx = 0.0f; sx = w - 1; }
    const int a0 = coef11((1.0f - fx) * 2048.0f), a1 = coef11(fx * 2048.0f);
    const int sx1 = min(sx + 1, w - 1);

    float fy = (float)(((double)y + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int b0 = coef11((1.0f - fy) * 2048.0f), b1 = coef11(fy * 2048.0f);
    const int sy0 = min(max(sy, 0), h - 1), sy1 = min(max(sy + 1, 0), h - 1);
    const uint8_t *S0 = src + (long long)sy0 * stride;
    const uint8_t *S1 = src + (long long)sy1 * stride;
    // The two taps of a row are the 6 consecutive bytes sx*3 .. sx*3+5 (sx1 = sx + 1): one unaligned 8-byte load per row instead
    // of six byte loads (round 3: the byte loads, not the arithmetic, held this kernel at 1.9 TB/s).  The wide load may touch
    // the 2 bytes behind the second tap, so it is used only where those still lie inside the frame's own extent; the last pixels
    // of the last row (and the clamped right edge, sx1 == sx) take the byte path.  Same integers either way.
    typedef uint64_t u64_unaligned __attribute__((aligned(1)));
    int t0[6], t1[6];
    // (the frame's real extent is (h-1) * stride + w * 3 bytes -- a caller's buffer with padded rows need not own the padding
    //  behind its LAST row (an OpenCV ROI, an exactly-sized allocation); round-3 advisor finding)
    const long long o0 = (long long)sy0 * stride + sx * 3, o1 = (long long)sy1 * stride + sx * 3, lim = (long long)(h - 1) * stride + (long long)w * 3 - 8;
    if (sx1 == sx + 1 && o0 <= lim && o1 <= lim) {
        const uint64_t q0 = *reinterpret_cast<const u64_unaligned *>(src + o0), q1 = *reinterpret_cast<const u64_unaligned *>(src + o1);
#pragma unroll
        for (int k = 0; k < 6; ++k) { t0[k] = (int)((q0 >> (8 * k)) & 0xff); t1[k] = (int)((q1 >> (8 * k)) & 0xff); }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { t0[c] = S0[sx * 3 + c]; t0[3 + c] = S0[sx1 * 3 + c]; t1[c] = S1[sx * 3 + c]; t1[3 + c] = S1[sx1 * 3 + c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int r0 = t0[c] * a0 + t0[3 + c] * a1;
        const int r1 = t1[c] * a0 + t1[3 + c] * a1;
        v[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v[c] = min(255, max(0, v[c]));
    }
}

__global__ void __launch_bounds__(256) preprocess_kernel(PreParams p)
{
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.net_w || y >= p.net_h) return;
    const PreImage im = p.imgs[b];

    int v[3] = {0, 0, 0}; // B,G,R of the canvas pixel; outside the pasted region the canvas is 0
    if (x < im.new_w && y < im.new_h) resize_linear_px(im.src, im.stride, im.h, im.w, im.scale_x, im.scale_y, im.area_fast, x, y, v);
    const size_t pix = ((size_t)b * p.net_h + y) * p.net_w + x;
    if (p.out_nhwc4) {
        // u8 -> bf16 is exact (8 significant bits); channel order R,G,B,0 (face_detection.rs:226 [2-i])
        const uint32_t r = __float_as_uint((float)v[2]) >> 16, g = __float_as_uint((float)v[1]) >> 16;
        const uint32_t bl = __float_as_uint((float)v[0]) >> 16;
        uint2 o;
        o.x = r | (g << 16);
        o.y = bl;
        *reinterpret_cast<uint2 *>(p.out_nhwc4 + pix * 4) = o;
    }
    if (p.out_det_img) {
        uint8_t *d = p.out_det_img + pix * 3;
        d[0] = (uint8_t)v[0]; d[1] = (uint8_t)v[1]; d[2] = (uint8_t)v[2];
    }
    if (p.out_tensor) {
        const size_t plane = (size_t)p.net_h * p.net_w;
        float *t = p.out_tensor + (size_t)b * 3 * plane + (size_t)y * p.net_w + x;
        // (p / pixel_scale - mean) / std with scale 1, mean 0, std 1 (face_detection.rs:105-107, 227)
        t[0] = ((float)v[2] / 1.0f - 0.0f) / 1.0f;
        t[plane] = ((float)v[1] / 1.0f - 0.0f) / 1.0f;
        t[2 * plane] = ((float)v[0] / 1.0f - 0.0f) / 1.0f;
    }
}

int launch_preprocess(const PreParams &p, int n, hipStream_t s)
{
    dim3 grid(ceil_div(p.net_w, 64), ceil_div(p.net_h, 4), n);
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

__global__ void __launch_bounds__(256) tensor_to_nhwc4_kernel(const float *__restrict__ t,
                                                              bf16_t *__restrict__ out, size_t plane,
                                                              size_t total)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; // pixel index over n*H*W
    if (i >= total) return;
    const size_t b = i / plane, q = i - b * plane;
    const float *s = t + b * 3 * plane + q;
    uint2 o;
    o.x = f32_to_bf16_bits(s[0]) | (f32_to_bf16_bits(s[plane]) << 16);
    o.y = f32_to_bf16_bits(s[2 * plane]);
    *reinterpret_cast<uint2 *>(out + i * 4) = o;
}

int launch_tensor_to_nhwc4(const float *tensor, bf16_t *out, int n, int H, int W, hipStream_t s)
{
    const size_t plane = (size_t)H * W, total = plane * n;
    hipLaunchKernelGGL(tensor_to_nhwc4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       tensor, out, plane, total);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

// ------------------------------------------------------------------------------------------------
// FaceAlignment::call (face_alignment.rs:27-141).  Set-up: one thread per face estimates the 4-DOF similarity from
// the five key points to the template as cv::estimateAffinePartial2D(LMEDS, 3.0, 2000, 0.99, 10) does (:48-60; restated
// from OpenCV 4.x calib3d ptsetreg.cpp as in oracle/rfd_oracle.c, where every step is cited): 13 two-point samples from a
// cv::RNG re-seeded on every call (so the SAME index pairs for every face), the exact similarity through each pair in f64,
// f32 squared errors, least median, inliers within 2.5 * 1.4826 * (1 + 5 / 3) * sqrt(median), then the least-squares
// similarity over the inliers (the fixed point of the reference's 10 Levenberg-Marquardt iterations on this linear problem;
// documented divergence ~1e-12).  Rounds 1-3 used the all-points closed form: no outlier rejection.  The matrix is then
// inverted as cv::warpAffine does; no model (coincident points) takes the reference's crop + resize branch (:62-110,
// quirks kept).
// Warp: one thread per output pixel, cv::warpAffine's fixed-point coordinates (10 -> 5 fractional bits) and 15-bit
// bilinear weights, out-of-image taps = 0.  Integer-exact against the oracle.
// ------------------------------------------------------------------------------------------------
constexpr int kLmedsIters = 13; // RANSACUpdateNumIters(0.99, 0.45, 2, 2000) = round(log(0.01) / log(1 - 0.55^2)); tests/test_oracle_cpu.py

__device__ __forceinline__ unsigned cv_rng_next(unsigned long long &state) // cv::RNG: multiply-with-carry
{
    state = (unsigned long long)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
    return (unsigned)state;
}

__device__ void similarity_2pt(const float *src, const float *dst, int i0, int i1, double M[6])
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

__device__ __forceinline__ void similarity_errors5(const float *src, const float *dst, const double M[6], float err[5])
{
    const float F0 = (float)M[0], F1 = (float)M[1], F2 = (float)M[2], F3 = (float)M[3], F4 = (float)M[4], F5 = (float)M[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float fx = src[2 * i], fy = src[2 * i + 1];
        const float a = F0 * fx + F1 * fy + F2 - dst[2 * i];
        const float b = F3 * fx + F4 * fy + F5 - dst[2 * i + 1];
        err[i] = a * a + b * b;
    }
}

// returns false: no model (the reference's empty matrix).  Every array is indexed at compile time (no scratch: tests/test_build_cpu.py).
__device__ bool estimate_similarity_lmeds5(const float *src, const float *dst, double M[6])
{
    unsigned long long rng = 0xffffffffffffffffull; // RNG rng((uint64)-1)
    double best[6] = {0, 0, 0, 0, 0, 0}, min_median = 1.7976931348623157e308;
    float err[5];
    for (int it = 0; it < kLmedsIters; ++it) {
        const int i0 = (int)(cv_rng_next(rng) % 5u);
        int i1;
        do i1 = (int)(cv_rng_next(rng) % 5u); while (i1 == i0);
        double Ms[6];
        similarity_2pt(src, dst, i0, i1, Ms);
        similarity_errors5(src, dst, Ms, err);
        // element 2 of the errors ordered by their bit patterns as ints (std::nth_element on errf.ptr<int>(): NaN above every number)
        int e0 = __float_as_int(err[0]), e1 = __float_as_int(err[1]), e2 = __float_as_int(err[2]), e3 = __float_as_int(err[3]), e4 = __float_as_int(err[4]);
#define RFD_CS(a, b) { const int lo_ = min(a, b), hi_ = max(a, b); a = lo_; b = hi_; }
        RFD_CS(e0, e1) RFD_CS(e3, e4) RFD_CS(e2, e4) RFD_CS(e2, e3) RFD_CS(e1, e4) RFD_CS(e0, e3) RFD_CS(e0, e2) RFD_CS(e1, e3) RFD_CS(e1, e2)
#undef RFD_CS
        const double median = (double)__int_as_float(e2);
        if (median < min_median) {
            min_median = median;
#pragma unroll
            for (int k = 0; k < 6; ++k) best[k] = Ms[k];
        }
    }
    if (!(min_median < 1.7976931348623157e308)) return false;
    double sigma = 2.5 * 1.4826 * (1 + 5. / 3) * sqrt(min_median);
    if (!(sigma > 0.001)) sigma = 0.001;
    const float t = (float)(sigma * sigma);
    similarity_errors5(src, dst, best, err);
    int cnt = 0;
    bool use[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { use[i] = err[i] <= t; cnt += use[i]; }
    if (cnt < 2) return false;
#pragma unroll
    for (int k = 0; k < 6; ++k) M[k] = best[k];
    // least squares over the inliers
    double msx = 0, msy = 0, mdx = 0, mdy = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (!use[i]) continue;
        msx += (double)src[2 * i]; msy += (double)src[2 * i + 1];
        mdx += (double)dst[2 * i]; mdy += (double)dst[2 * i + 1];
    }
    msx /= cnt; msy /= cnt; mdx /= cnt; mdy /= cnt;
    double sxx = 0, sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (!use[i]) continue;
        const double xs = (double)src[2 * i] - msx, ys = (double)src[2 * i + 1] - msy;
        const double xd = (double)dst[2 * i] - mdx, yd = (double)dst[2 * i + 1] - mdy;
        sxx += xs * xs + ys * ys;
        sa += xs * xd + ys * yd;
        sb += xs * yd - ys * xd;
    }
    if (sxx > 0.0) {
        const double a = sa / sxx, bb = sb / sxx;
        M[0] = a; M[1] = -bb; M[2] = mdx - (a * msx - bb * msy);
        M[3] = bb; M[4] = a; M[5] = mdy - (bb * msx + a * msy);
    }
    return true;
}

__device__ __forceinline__ int cv_round_sat(double v)
{
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (-2147483647 - 1);
    return __double2int_rn(v);
}

__global__ void align_setup_kernel(AlignParams p)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.n) return;
    AlignFace f;
    memset(&f, 0, sizeof f);
    const int found = p.found[b];
    const PreImage im = p.imgs[b];
    if (!(found & 1)) {
        f.mode = -2;
    } else if (!(found & 2)) {
        f.mode = -1; // landmarks = None: estimate_affine_partial_2d rejects the empty point set (:48)
    } else {
        const float *src = p.kps + (size_t)b * 10;
        double M[6];
        if (estimate_similarity_lmeds5(src, p.std_lmk, M)) {
            double D = M[0] * M[4] - M[1] * M[3];
            D = D != 0 ? 1. / D : 0;
            const double A11 = M[4] * D, A22 = M[0] * D;
            M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
            const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
            M[2] = b1; M[5] = b2;
            for (int i = 0; i < 6; ++i) f.M[i] = M[i];
            f.mode = 0;
        } else { // empty transformation (:62): crop around the box and resize
            const float *det = p.box + (size_t)b * 5;
            const float margin = 44.0f;
            const float bb0 = fmaxf(det[0] - margin / 2.0f, 0.0f), bb1 = fmaxf(det[1] - margin / 2.0f, 0.0f);
            const float bb2 = fmaxf(det[2] + margin / 2.0f, (float)im.w); // `max`, as written (:77)
            const float bb3 = fmaxf(det[1] + margin / 2.0f, (float)im.h); // det[1], as written (:78)
            f.x0 = (int)bb0; f.y0 = (int)bb1;
            f.rw = (int)bb2 - f.x0; f.rh = (int)bb3 - f.y0;
            if (f.rw <= 0 || f.rh <= 0 || f.x0 + f.rw > im.w || f.y0 + f.rh > im.h) {
                f.mode = -3;
            } else {
                f.mode = 1;
                f.scale_x = 1.0 / ((double)p.out_w / f.rw);
                f.scale_y = 1.0 / ((double)p.out_h / f.rh);
                const int ix = cv_round_sat(f.scale_x), iy = cv_round_sat(f.scale_y);
                f.area_fast = fabs(f.scale_x - ix) < 2.220446049250313e-16 && fabs(f.scale_y - iy) < 2.220446049250313e-16 &&
                              ix == 2 && iy == 2;
            }
        }
    }
    p.faces[b] = f;
    p.status[b] = f.mode;
}

__global__ void __launch_bounds__(256) align_warp_kernel(AlignParams p)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.out_w * p.out_h) return;
    const int y = i / p.out_w, x = i - y * p.out_w;
    const AlignFace &f = p.faces[b];
    uint8_t *d = p.out + ((size_t)b * p.out_h * p.out_w + i) * 3;
    if (f.mode < 0) { d[0] = d[1] = d[2] = 0; return; }
    const PreImage im = p.imgs[b];
    int v[3];
    if (f.mode == 1) {
        resize_linear_px(im.src + (long long)f.y0 * im.stride + f.x0 * 3, im.stride, f.rh, f.rw, f.scale_x, f.scale_y,
                         f.area_fast, x, y, v);
    } else {
        constexpr int AB_BITS = 10, AB_SCALE = 1 << AB_BITS, INTER_BITS = 5, TAB = 1 << INTER_BITS;
        constexpr int round_delta = AB_SCALE / TAB / 2;
        const int X0 = cv_round_sat((f.M[1] * y + f.M[2]) * AB_SCALE) + round_delta;
        const int Y0 = cv_round_sat((f.M[4] * y + f.M[5]) * AB_SCALE) + round_delta;
        const int adelta = cv_round_sat(f.M[0] * x * AB_SCALE), bdelta = cv_round_sat(f.M[3] * x * AB_SCALE);
        const int X = (X0 + adelta) >> (AB_BITS - INTER_BITS), Y = (Y0 + bdelta) >> (AB_BITS - INTER_BITS);
        const int sx = max(-32768, min(32767, X >> INTER_BITS)), sy = max(-32768, min(32767, Y >> INTER_BITS));
        const int fx = X & (TAB - 1), fy = Y & (TAB - 1);
        int wt[4] = {(TAB - fy) * (TAB - fx) * 32, (TAB - fy) * fx * 32, fy * (TAB - fx) * 32, fy * fx * 32};
        if (fx == 0 && fy == 0) { wt[0] = 32767; wt[3] = 1; }
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int px = sx + (t & 1), py = sy + (t >> 1);
            if (px >= 0 && px < im.w && py >= 0 && py < im.h) {
                const uint8_t *s = im.src + (long long)py * im.stride + px * 3;
                acc[0] += s[0] * wt[t]; acc[1] += s[1] * wt[t]; acc[2] += s[2] * wt[t];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (acc[c] + (1 << 14)) >> 15;
    }
    d[0] = (uint8_t)v[0]; d[1] = (uint8_t)v[1]; d[2] = (uint8_t)v[2];
}

int launch_face_align(const AlignParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(align_setup_kernel, dim3(ceil_div(p.n, 64)), dim3(64), 0, s, p);
    RFD_HIP(hipGetLastError());
    hipLaunchKernelGGL(align_warp_kernel, dim3(ceil_div(p.out_w * p.out_h, 256), p.n), dim3(256), 0, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

} // namespace rfd

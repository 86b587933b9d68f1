// kernels_f32.hip -- the f32 PARITY MODE of the network (rfd_config.precision = RFD_PRECISION_F32).
//
// The reference's tensor contract is FP32 end to end (face_detection.rs:261 "FP32", :270 fp32_contents): an f32 image tensor goes
// to an f32 Triton model and f32 heads come back.  The product path computes in bf16 (BASELINE.json configs[2] names bf16), which
// moves scores by ~3e-2 and boxes by up to a few pixels against an f32 evaluation of the same weights (tests/test_t2_gpu.py).
// This file is the mode in which north_star's "identical kept-box index sets, coordinates within 1e-4" can be shown END TO END
// against an independent f32 evaluation: the same graph (same fusions, same folded parameters, same op list) with f32 weights,
// f32 activations and -- since round 4 -- f64 ACCUMULATION with one rounding per convolution output: products of two f32 values are
// exact in f64, so every conv output is the correctly rounded value of its exact sum (up to ~K * 2^-53, i.e. one output in ~10^5
// lands on the other side of an f32 rounding boundary), whatever order the terms are added in.  Two f32 evaluations with
// different summation orders differ by 1-3e-6 in the head tensors, which a 512-pixel anchor turns into 7e-4 px (round 3); two
// f64-accumulating ones agree to the ulp, and north_star's 1e-4 becomes checkable end to end (tests/test_t2_gpu.py against
// tests/torch_ref.py with acc64=True: f64 conv, .float() per layer).  Everything outside the sums is plain f32 in a fixed order.
// It is a correctness mode, not a fast path: plain LDS-tiled FMA kernels (64 x 64 outputs per workgroup, K step 16), no MFMA,
// no persistent kernels, one stream; ~60x slower than the bf16 path.
//
// Semantics follow the bf16 kernels op for op (kernels_conv.hip: conv_epilogue, stem_kernel, conv_b2b_s1_kernel) minus every
// rounding of a stored tensor; the network input stays the bf16 NHWC4 tensor the preprocess kernel writes (raw 0..255: exact).
#include "kernels.h"

namespace rfd {

namespace {

constexpr int kTM = 64, kTN = 64, kTK = 16;

__global__ void __launch_bounds__(256) conv_f32_kernel(const ConvF32Params p)
{
    __shared__ float Xs[kTK][kTM + 4];
    __shared__ float Ws[kTK][kTN + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int HoWo = p.Ho * p.Wo, M = p.B * HoWo;
    const int K1 = p.KH * p.KW * p.Cin, K = K1 + p.Cin2;
    const int tiles_n = (p.Cout + kTN - 1) / kTN;
    const int m0 = (blockIdx.x / tiles_n) * kTM, n0 = (blockIdx.x % tiles_n) * kTN;

    // loader role: this thread stages pixel (tid / 4) and weight row (tid / 4), K quad (tid % 4) of every K step
    const int lrow = tid >> 2, lq = (tid & 3) * 4;
    const int lm = m0 + lrow;
    int lb = 0, lho = 0, lwo = 0;
    const bool lm_ok = lm < M;
    if (lm_ok) { lb = lm / HoWo; const int rem = lm - lb * HoWo; lho = rem / p.Wo; lwo = rem - lho * p.Wo; }
    const int ln = n0 + lrow;
    const bool ln_ok = ln < p.Cout;

    double acc[4][4]; // f64: the products are exact, the sum is rounded to f32 once (below)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;

    for (int k0 = 0; k0 < K; k0 += kTK) {
        // ---- stage X[kTM][kTK] (im2col) and W[kTN][kTK]; Cin, Cin2 are multiples of 16, so a K step never straddles a tap ----
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = k0 + lq;
        if (lm_ok) {
            if (k < K1) {
                const int tap = k / p.Cin, ci = k - tap * p.Cin, ky = tap / p.KW, kx = tap - ky * p.KW;
                const int hi = lho * p.stride - p.pad + ky, wi = lwo * p.stride - p.pad + kx;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                    xv = *reinterpret_cast<const float4 *>(p.x + (((size_t)lb * p.H + hi) * p.W + wi) * p.ldx + p.x_coff + ci);
                    if (p.in_scale) { // BN+ReLU of the producer unit on the operand; padding taps stay exactly 0
                        const float4 s = *reinterpret_cast<const float4 *>(p.in_scale + ci), t = *reinterpret_cast<const float4 *>(p.in_shift + ci);
                        xv.x = fmaxf(fmaf(xv.x, s.x, t.x), 0.f); xv.y = fmaxf(fmaf(xv.y, s.y, t.y), 0.f);
                        xv.z = fmaxf(fmaf(xv.z, s.z, t.z), 0.f); xv.w = fmaxf(fmaf(xv.w, s.w, t.w), 0.f);
                    }
                }
            } else { // second K segment: the fused 1x1 shortcut conv (stride2, no pad) on x2
                const int ci = k - K1;
                xv = *reinterpret_cast<const float4 *>(p.x2 + (((size_t)lb * p.H2 + lho * p.stride2) * p.W2 + lwo * p.stride2) * p.Cin2 + ci);
            }
        }
        float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ln_ok) wv = *reinterpret_cast<const float4 *>(p.w + (size_t)ln * p.ldw + k);
        __syncthreads(); // the previous step's tiles have been consumed
        Xs[lq + 0][lrow] = xv.x; Xs[lq + 1][lrow] = xv.y; Xs[lq + 2][lrow] = xv.z; Xs[lq + 3][lrow] = xv.w;
        Ws[lq + 0][lrow] = wv.x; Ws[lq + 1][lrow] = wv.y; Ws[lq + 2][lrow] = wv.z; Ws[lq + 3][lrow] = wv.w;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kTK; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = (double)Xs[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = (double)Ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
    }

    // ---- epilogue: the op semantics of conv_epilogue (kernels_conv.hip), without the bf16 roundings ----
    const int nb = n0 + tx * 4;
    if (nb >= p.Cout) return;
    float bias[4], s2[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bias[j] = p.bias[nb + j] + (p.bias2 ? p.bias2[nb + j] : 0.f);
        if (p.y2) { s2[j] = p.scale2[nb + j]; t2[j] = p.shift2[nb + j]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 16 * i;
        if (m >= M) continue;
        float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)acc[i][j] + bias[j]; // the one rounding of the sum, then f32 in a fixed order
        if (p.res) {
            size_t mr = (size_t)m;
            if (p.res_up2) {
                const int b = m / HoWo, rem = m - b * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                mr = ((size_t)b * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1);
            }
            const float4 rv = *reinterpret_cast<const float4 *>(p.res + mr * p.Cout + nb);
            r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w;
            if (!p.res_post) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += r[j];
            }
        }
        if (p.y && nb < p.n_valid) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = p.relu ? fmaxf(v[j], 0.f) : v[j];
                if (p.res && p.res_post) o[j] += r[j];
            }
            const int nd = nb + p.y_coff + (nb >= p.y_split ? p.y_split_add : 0);
            *reinterpret_cast<float4 *>(p.y + (size_t)m * p.ldy + nd) = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (p.y2) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(v[j] * s2[j] + t2[j], 0.f);
            *reinterpret_cast<float4 *>(p.y2 + (size_t)m * p.Cout + nb) = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (p.yf) {
            if (p.head_softmax && nb == 0) { // channels 0..3 = bg0,bg1,fg0,fg1: 2-class softmax over the pairs (a, A+a)
                const float m0s = fmaxf(v[0], v[2]), m1s = fmaxf(v[1], v[3]);
                const float e0 = expf(v[0] - m0s), e2 = expf(v[2] - m0s);
                const float e1 = expf(v[1] - m1s), e3 = expf(v[3] - m1s);
                v[0] = e0 / (e0 + e2); v[2] = e2 / (e0 + e2);
                v[1] = e1 / (e1 + e3); v[3] = e3 / (e1 + e3);
            }
            *reinterpret_cast<float4 *>(p.yf + (size_t)m * p.Cout + nb) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// conv0: 7x7 / stride 2 / pad 3 on the bf16 NHWC4 input (R,G,B,0), weights f32 [64][7][8 kx][4 c], + bias + ReLU -> f32 [B][H/2][W/2][64]
__global__ void __launch_bounds__(256) conv0_f32_kernel(const bf16_t *__restrict__ x4, const float *__restrict__ w,
                                                        const float *__restrict__ bias, float *__restrict__ y, int B, int H, int W)
{
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)B * Ho * Wo * 16; // 4 output channels per thread
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int n4 = (int)(t & 15) * 4;
    const long long m = t >> 4;
    const int b = (int)(m / ((long long)Ho * Wo)), rem = (int)(m - (long long)b * Ho * Wo), ho = rem / Wo, wo = rem - ho * Wo;
    double acc[4] = {0.0, 0.0, 0.0, 0.0}; // f64 accumulation, one rounding (see the file header)
    for (int ky = 0; ky < 7; ++ky) {
        const int hi = 2 * ho - 3 + ky;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int kx = 0; kx < 7; ++kx) {
            const int wi = 2 * wo - 3 + kx;
            if ((unsigned)wi >= (unsigned)W) continue;
            const uint2 px = *reinterpret_cast<const uint2 *>(x4 + (((size_t)b * H + hi) * W + wi) * 4);
            const float c0 = __uint_as_float(px.x << 16), c1 = __uint_as_float(px.x & 0xffff0000u), c2 = __uint_as_float(px.y << 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float *wr = w + ((size_t)(n4 + j) * 7 + ky) * 32 + kx * 4;
                acc[j] = fma((double)c0, (double)wr[0], acc[j]);
                acc[j] = fma((double)c1, (double)wr[1], acc[j]);
                acc[j] = fma((double)c2, (double)wr[2], acc[j]);
            }
        }
    }
    *reinterpret_cast<float4 *>(y + (size_t)m * 64 + n4) =
        make_float4(fmaxf((float)acc[0] + bias[n4], 0.f), fmaxf((float)acc[1] + bias[n4 + 1], 0.f), fmaxf((float)acc[2] + bias[n4 + 2], 0.f),
                    fmaxf((float)acc[3] + bias[n4 + 3], 0.f));
}

// 3x3 / stride 2 / pad 1 max pool, then per-channel affine + ReLU (the BN1+ReLU that opens the first unit); 4 channels per thread
__global__ void __launch_bounds__(256) maxpool_f32_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int B, int H, int W, int C, int Ho, int Wo)
{
    const int c4n = C >> 2;
    const long long total = (long long)B * Ho * Wo * c4n;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % c4n) * 4;
    const long long m = t / c4n;
    const int b = (int)(m / ((long long)Ho * Wo)), rem = (int)(m - (long long)b * Ho * Wo), ho = rem / Wo, wo = rem - ho * Wo;
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dy = 0; dy < 3; ++dy) {
        const int hi = 2 * ho - 1 + dy;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int dx = 0; dx < 3; ++dx) {
            const int wi = 2 * wo - 1 + dx;
            if ((unsigned)wi >= (unsigned)W) continue;
            const float4 v = *reinterpret_cast<const float4 *>(x + (((size_t)b * H + hi) * W + wi) * C + c);
            mx[0] = fmaxf(mx[0], v.x); mx[1] = fmaxf(mx[1], v.y); mx[2] = fmaxf(mx[2], v.z); mx[3] = fmaxf(mx[3], v.w);
        }
    }
    const float4 s = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
    *reinterpret_cast<float4 *>(y + (size_t)m * C + c) =
        make_float4(fmaxf(mx[0] * s.x + sh.x, 0.f), fmaxf(mx[1] * s.y + sh.y, 0.f), fmaxf(mx[2] * s.z + sh.z, 0.f), fmaxf(mx[3] * s.w + sh.w, 0.f));
}

} // namespace

int launch_conv_f32(const ConvF32Params &p, hipStream_t s)
{
    if (p.Cin % kTK != 0 || p.Cin2 % kTK != 0 || p.Cout % 4 != 0 || p.ldx % 4 != 0 || p.x_coff % 4 != 0 || p.ldw % 4 != 0) {
        set_error("f32 conv: unsupported shape Cin=%d Cin2=%d Cout=%d", p.Cin, p.Cin2, p.Cout);
        return RFD_ERR_INVALID_ARG;
    }
    const int M = p.B * p.Ho * p.Wo;
    const long long grid = (long long)ceil_div(M, kTM) * ceil_div(p.Cout, kTN);
    hipLaunchKernelGGL(conv_f32_kernel, dim3((unsigned)grid), dim3(256), 0, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

int launch_conv0_f32(const bf16_t *x4, const float *w, const float *bias, float *y, int B, int H, int W, hipStream_t s)
{
    const long long total = (long long)B * (H / 2) * (W / 2) * 16;
    hipLaunchKernelGGL(conv0_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x4, w, bias, y, B, H, W);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

int launch_maxpool_f32(const float *x, float *y, const float *scale, const float *shift, int B, int H, int W, int C, hipStream_t s)
{
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, scale, shift, B, H, W, C, Ho, Wo);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

} // namespace rfd

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

__global__ void __launch_bounds__(256) preprocess_kernel(PreParams p)
{
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= p.net_w || y >= p.net_h) return;
    const PreImage im = p.imgs[b];

    int v[3] = {0, 0, 0}; // B,G,R of the canvas pixel; outside the pasted region the canvas is 0
    if (x < im.new_w && y < im.new_h) {
        if (im.area_fast) {
            const uint8_t *s0 = im.src + (long long)(2 * y) * im.stride + 2 * x * 3;
            const uint8_t *s1 = s0 + im.stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (s0[c] + s0[c + 3] + s1[c] + s1[c + 3] + 2) >> 2;
        } else {
            float fx = (float)(((double)x + 0.5) * im.scale_x - 0.5);
            int sx = (int)floorf(fx);
            fx -= (float)sx;
            if (sx < 0) { fx = 0.0f; sx = 0; }
            if (sx >= im.w - 1) { fx = 0.0f; sx = im.w - 1; }
            const int a0 = coef11((1.0f - fx) * 2048.0f), a1 = coef11(fx * 2048.0f);
            const int sx1 = min(sx + 1, im.w - 1);

            float fy = (float)(((double)y + 0.5) * im.scale_y - 0.5);
            const int sy = (int)floorf(fy);
            fy -= (float)sy;
            const int b0 = coef11((1.0f - fy) * 2048.0f), b1 = coef11(fy * 2048.0f);
            const int sy0 = min(max(sy, 0), im.h - 1), sy1 = min(max(sy + 1, 0), im.h - 1);
            const uint8_t *S0 = im.src + (long long)sy0 * im.stride;
            const uint8_t *S1 = im.src + (long long)sy1 * im.stride;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r0 = S0[sx * 3 + c] * a0 + S0[sx1 * 3 + c] * a1;
                const int r1 = S1[sx * 3 + c] * a0 + S1[sx1 * 3 + c] * a1;
                v[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                v[c] = min(255, max(0, v[c]));
            }
        }
    }
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

} // namespace rfd

// conv_device.h -- device helpers shared by the convolution translation units (kernels_conv.hip, kernels_ring.hip):
// MFMA operand types, the XCD-aware tile map, the LDS-DMA primitive, and the register epilogue of the implicit-GEMM kernels.
#pragma once
#include "kernels.h"

namespace rfd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d)
{
    bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return __builtin_bit_cast(uint2, v);
}

// blockIdx -> tile id such that each XCD (blocks b, b+8, ... share one L2) walks a contiguous chunk
// of tiles: neighbouring tiles share weight panels / activation halos (bijective for any grid size)
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// LDS-DMA: one wave instruction moves 64 x 16 B straight from global memory into LDS at
// (wave-uniform base) + lane*16; the per-lane SOURCE address carries the swizzle.
// Buffer form (buffer_load_dwordx4 ... offen lds): 32-bit per-lane byte offset + scalar offset, and the
// hardware range check returns ZEROS for a per-lane offset >= num_records -- conv padding for free.
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, uint32_t soffset,
                                       void *lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16,
                                             voffset, soffset, 0, 0);
}
constexpr uint32_t kOob = 0xfffffff0u; // >= any num_records: reads as zeros, touches no memory

// residual prefetch (independent of the GEMM): 16 bytes = the lane's 8 channels of pixel (j)
template <int TM, int TH, int WM, int WN>
__device__ __forceinline__ void conv_prefetch_residual(const ConvParams &p, uint4 (&resv)[TM][TH], int m0, int n0, int wm,
                                                       int wn, int frow, int fq, int M, int HoWo)
{
    if (!p.res) return;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * WM + j * 16 + frow;
        size_t mr = (size_t)(m < M ? m : 0);
        if (p.res_up2 && m < M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            mr = ((size_t)b * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1);
        }
#pragma unroll
        for (int h = 0; h < TH; ++h)
            resv[j][h] = *reinterpret_cast<const uint4 *>(p.res + mr * p.Cout + n0 + wn * WN + h * 32 + fq * 8);
    }
}

template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, f32x4 (&acc)[TN][TM], uint4 (&resv)[TM][TN / 2], int m0,
                                              int n0, int wm, int wn, int frow, int fq, int M)
{
    constexpr int TH = TN / 2;
    // ---- fused epilogue from registers: lane = pixel (j*16 + frow), channels h*32 + fq*8 .. +7 ----
#pragma unroll
    for (int h = 0; h < TH; ++h) {
        const int n = n0 + wn * WN + h * 32 + fq * 8;
        float bias[8], s2[8], t2[8];
        {
            const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + n), b1 = *reinterpret_cast<const float4 *>(p.bias + n + 4);
            bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w;
            bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
            if (p.bias2) {
                const float4 d0 = *reinterpret_cast<const float4 *>(p.bias2 + n), d1 = *reinterpret_cast<const float4 *>(p.bias2 + n + 4);
                bias[0] += d0.x; bias[1] += d0.y; bias[2] += d0.z; bias[3] += d0.w;
                bias[4] += d1.x; bias[5] += d1.y; bias[6] += d1.z; bias[7] += d1.w;
            }
        }
        if (p.y2) {
            const float4 a0 = *reinterpret_cast<const float4 *>(p.scale2 + n), a1 = *reinterpret_cast<const float4 *>(p.scale2 + n + 4);
            const float4 c0 = *reinterpret_cast<const float4 *>(p.shift2 + n), c1 = *reinterpret_cast<const float4 *>(p.shift2 + n + 4);
            s2[0] = a0.x; s2[1] = a0.y; s2[2] = a0.z; s2[3] = a0.w; s2[4] = a1.x; s2[5] = a1.y; s2[6] = a1.z; s2[7] = a1.w;
            t2[0] = c0.x; t2[1] = c0.y; t2[2] = c0.z; t2[3] = c0.w; t2[4] = c1.x; t2[5] = c1.y; t2[6] = c1.z; t2[7] = c1.w;
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * WM + j * 16 + frow;
            if (m >= M) continue;
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = acc[2 * h][j][k] + bias[k];
                v[4 + k] = acc[2 * h + 1][j][k] + bias[4 + k];
            }
            float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.res) {
                const uint4 rv = resv[j][h];
                r[0] = bf16_bits_to_f32(rv.x & 0xffffu); r[1] = bf16_bits_to_f32(rv.x >> 16);
                r[2] = bf16_bits_to_f32(rv.y & 0xffffu); r[3] = bf16_bits_to_f32(rv.y >> 16);
                r[4] = bf16_bits_to_f32(rv.z & 0xffffu); r[5] = bf16_bits_to_f32(rv.z >> 16);
                r[6] = bf16_bits_to_f32(rv.w & 0xffffu); r[7] = bf16_bits_to_f32(rv.w >> 16);
                if (!p.res_post) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += r[k];
                }
            }
            if (p.y && n < p.n_valid) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    o[k] = p.relu ? fmaxf(v[k], 0.f) : v[k];
                    if (p.res && p.res_post) o[k] += r[k];
                }
                const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                const int nd = n + p.y_coff + (n >= p.y_split ? p.y_split_add : 0);
                *reinterpret_cast<uint4 *>(p.y + (size_t)m * p.ldy + nd) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            if (p.y2) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = fmaxf(v[k] * s2[k] + t2[k], 0.f);
                const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                *reinterpret_cast<uint4 *>(p.y2 + (size_t)m * p.Cout + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            if (p.yf) {
                if (p.head_softmax && n == 0) {
                    // channels 0..3 = bg0,bg1,fg0,fg1: 2-class softmax over the pairs (a, A+a)
                    const float m0s = fmaxf(v[0], v[2]), m1s = fmaxf(v[1], v[3]);
                    const float e0 = expf(v[0] - m0s), e2 = expf(v[2] - m0s);
                    const float e1 = expf(v[1] - m1s), e3 = expf(v[3] - m1s);
                    v[0] = e0 / (e0 + e2); v[2] = e2 / (e0 + e2);
                    v[1] = e1 / (e1 + e3); v[3] = e3 / (e1 + e3);
                }
                float4 *dst = reinterpret_cast<float4 *>(p.yf + (size_t)m * p.Cout + n);
                dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
}

} // namespace rfd

// lds_dma_hazard.hip -- stand-alone reproducer of a cross-wave problem met on gfx950 (MI355X, ROCm 7.0 runtime).
//
// Symptom: the old MobileNet first-conv kernel kept its 288 weights as floats in LDS and read them with broadcast
// ds_read_b128 (all lanes, same address).  Whenever waves of ANOTHER kernel that issue MFMAs were resident on the same
// CU, some of its outputs were wrong -- always in lanes 48..63 of a wave, only in the channels fed by ds_read_b128.
//
// What this program establishes (run it: the table it prints is the evidence; numbers from one MI355X):
//   * victim = that kernel (victim_f3<0>, compiler-chosen reads) or the same kernel with explicit broadcast
//     ds_read_b128 and a full s_waitcnt lgkmcnt(0) before any arithmetic (victim_f3<3>): 10^4..10^6 wrong elements per
//     run, all in lane quarter 3;  the same kernel reading the table as 2 x ds_read_b64 (victim_f3<4>): ZERO;
//   * disturber = any kernel issuing v_mfma_f32_16x16x32_bf16: LDS-DMA + fragment reads + MFMA (the conv kernels'
//     loop shape), fragment reads + MFMA without DMA, and MFMA on register operands with no LDS traffic at all;
//     LDS-DMA or LDS traffic WITHOUT MFMA does not trigger it (the file name records the first, wrong suspicion);
//   * still wrong with the global loads taken out of the victim's loop (pixels made from coordinates); 20 x FEWER wrong
//     elements when every wave repeats its nine taps 32 times and keeps the last round -- the errors sit early in a
//     wave's life, i.e. around workgroup launch / the table fill + barrier, not in steady state;
//   * plain read-and-compare victims of every shape (broadcast / per-lane / fragment b128, b96, b64, b32, eight reads in
//     flight, with a global load pending, feeding v_mul_f32 / v_pk_mul_f32) stay clean under every disturber: those
//     waves live for thousands of reads, which fits the previous point, but the mechanism is not isolated.
// The product therefore keeps broadcast operands out of ds_read_b128: scalar loads for weight tables, 64-bit LDS reads
// or v_readlane in sort / NMS (tests/test_concurrency_gpu.py guards the pipeline).  The conv kernels' own per-lane
// fragment ds_read_b128 next to their own MFMAs are not affected (bit-identical results in every execution mode).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lds_dma_hazard tools/lds_dma_hazard.hip && ./lds_dma_hazard
//   hipcc ... -shared -fPIC -o tools/bin/liblds_dma_hazard.so tools/lds_dma_hazard.hip   (tools/hazard_with_conv.py:
//   the same victims with the real conv kernels of librfd_hip.so as the disturber)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- disturbers ----
// mode 0: LDS-DMA dwordx4; mode 1: LDS-DMA dword; mode 2: global_load + ds_write_b128 (no DMA)
template <int MODE> __global__ void __launch_bounds__(256) disturber(const uint32_t *src, size_t n_bytes, int iters, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, (uint32_t)n_bytes, 0x00020000);
    uint32_t acc = 0;
    typedef __attribute__((ext_vector_type(4))) float macc_t;
    macc_t macc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) macc[a][b] = macc_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const uint32_t base = (uint32_t)(((size_t)(blockIdx.x * 131 + it) * 4096) % (n_bytes - 65536));
        for (int q = 0; q < 8; ++q) {
            unsigned char *dst = smem + (wave * 8 + q) * 1024;
            if (MODE == 0 || MODE == 3 || MODE == 4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)dst, 16,
                                                         base + (wave * 8 + q) * 1024 + lane * 16, 0, 0, 0);
            } else if (MODE == 1) {
                for (int k = 0; k < 4; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(dst + k * 256), 4,
                                                             base + (wave * 8 + q) * 1024 + k * 256 + lane * 4, 0, 0, 0);
            } else if (MODE == 2) {
                const uint4 v = *reinterpret_cast<const uint4 *>((const char *)src + base + (wave * 8 + q) * 1024 + lane * 16);
                *reinterpret_cast<uint4 *>(dst + lane * 16) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<const uint32_t *>(smem + ((threadIdx.x * 52 + it * 4) & 32764));
        if (MODE >= 3) { // the conv kernels' inner loop shape: 16 fragment ds_read_b128 (+ 32 MFMAs in mode 4) per step
            typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
            typedef __attribute__((ext_vector_type(4))) float f4;
            const int frow = lane & 15, fq = lane >> 4;
            bf8 af[4], bf[4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = t * 16 + frow, ch = kk * 4 + fq;
                    if (MODE == 6 && it > 0) continue; // MFMA on register operands only: the fragments are read once
                    af[t] = *reinterpret_cast<const bf8 *>(smem + (wave * 4096) + r * 128 + ((ch ^ (r & 7)) << 4));
                    bf[t] = *reinterpret_cast<const bf8 *>(smem + 16384 + (wave * 4096) + r * 128 + ((ch ^ (r & 7)) << 4));
                }
                if (MODE >= 4) {
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) macc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], macc[a][b], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc += __builtin_bit_cast(uint4, af[t]).x ^ __builtin_bit_cast(uint4, bf[t]).y;
                }
            }
        }
        __syncthreads();
    }
    if (MODE >= 4) {
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) t += macc[a][b][0] + macc[a][b][3];
        if (t == 123.456f) sink[1] = 1;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// ---- victims: fill a 4 KiB LDS table with a known pattern, then read it back many times ----
__device__ __forceinline__ uint32_t pat(uint32_t i) { return i * 2654435761u + 12345u; }

// SHAPE 0: broadcast b128; 1: per-lane b128 (lane*16); 2: broadcast b64; 3: broadcast b32; 4: broadcast b96;
//       5: MFMA-fragment-like b128 ((lane&15)*64 + (lane>>4)*16); 6: per-lane b64; 7: half-broadcast b128 (lane>>5)
template <int SHAPE> __global__ void __launch_bounds__(256) victim(int iters, unsigned long long *bad_by_lane)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = pat(i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int row = (it * 7 + (threadIdx.x >> 6)) & 31; // 32 rows of 128 B, wave-uniform
        int off; // dword offset
        if (SHAPE == 0 || SHAPE == 2 || SHAPE == 3 || SHAPE == 4) off = row * 32 + ((it & 7) * 4);
        else if (SHAPE == 1) off = ((row & 3) * 256 + lane * 4);
        else if (SHAPE == 5) off = ((lane & 15) * 16 + (lane >> 4) * 4 + (row & 3) * 256);
        else if (SHAPE == 6) off = ((row & 7) * 128 + lane * 2);
        else off = row * 32 + (lane >> 5) * 4;
        uint32_t v[4] = {0, 0, 0, 0};
        int n = 4;
        if (SHAPE == 2 || SHAPE == 6) {
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(__attribute__((ext_vector_type(2))) uint32_t *)v) : "v"(off * 4) : "memory");
            n = 2;
        } else if (SHAPE == 3) {
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v[0]) : "v"(off * 4) : "memory");
            n = 1;
        } else if (SHAPE == 4) {
            asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(__attribute__((ext_vector_type(3))) uint32_t *)v) : "v"(off * 4) : "memory");
            n = 3;
        } else {
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(__attribute__((ext_vector_type(4))) uint32_t *)v) : "v"(off * 4) : "memory");
        }
        for (int k = 0; k < n; ++k) bad += v[k] != pat(off + k);
    }
    if (bad) atomicAdd(&bad_by_lane[lane], (unsigned long long)bad);
}

// ---- round 2: the minimal form.  Plain read-and-compare of a 4 KiB table, but the read executes under a PARTIAL EXEC
//      mask: the lanes selected by `off_mask` (bit i = lane i) skip the read through a divergent branch, exactly what
//      `if (tap outside the image) continue;` did to lane 0 of every fifth wave in the first-conv kernel.
//      WIDTH 128 / 64: ds_read_b128 / 2 x ds_read_b64; ADDR 0 broadcast, 1 per-lane (lane*16), 2 MFMA-fragment pattern ----
template <int WIDTH, int ADDR> __global__ void __launch_bounds__(256) victim_masked(int iters, unsigned long long off_mask, unsigned long long *bad_by_lane)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = pat(i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool skip = (off_mask >> lane) & 1ull;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int row = (it * 7 + (threadIdx.x >> 6)) & 31;
        int off; // dwords
        if (ADDR == 0) off = row * 32 + ((it & 7) * 4);
        else if (ADDR == 1) off = ((row & 3) * 256 + lane * 4);
        else off = ((lane & 15) * 16 + (lane >> 4) * 4 + (row & 3) * 256);
        if (skip) continue; // divergent: the reads below run with EXEC = ~off_mask
        uint32_t v[4];
        if (WIDTH == 128) {
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(*(__attribute__((ext_vector_type(4))) uint32_t *)v) : "v"(off * 4) : "memory");
        } else {
            asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(*(__attribute__((ext_vector_type(2))) uint32_t *)v), "=&v"(*(__attribute__((ext_vector_type(2))) uint32_t *)(v + 2)) : "v"(off * 4) : "memory");
        }
        for (int k = 0; k < 4; ++k) bad += v[k] != pat(off + k);
    }
    if (bad) atomicAdd(&bad_by_lane[lane], (unsigned long long)bad);
}

// SHAPE 8/9: eight LDS reads in flight (4 x b128 + 4 x b96, broadcast) with (9) or without (8) a global load pending,
// the instruction pattern of the first-conv kernel that exposed the problem
template <int WITH_VMEM> __global__ void __launch_bounds__(256) victim_multi(int iters, unsigned long long *bad_by_lane, const uint32_t *g)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = pat(i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned bad = 0;
    uint32_t gsum = 0;
    for (int it = 0; it < iters; ++it) {
        const int off = ((it * 7 + (threadIdx.x >> 6)) & 15) * 36; // dwords; 144-byte records like the weight table
        typedef __attribute__((ext_vector_type(4))) uint32_t u4;
        typedef __attribute__((ext_vector_type(3))) uint32_t u3;
        u4 a0, a1, a2, a3;
        u3 b0, b1, b2, b3;
        uint32_t gv = 0;
        if (WITH_VMEM) gv = g[(blockIdx.x * 256 + threadIdx.x + it * 64) & 0xfffff];
        asm volatile("ds_read_b128 %0, %8\n ds_read_b96 %4, %8 offset:144\n ds_read_b96 %5, %8 offset:864\n ds_read_b96 %6, %8 offset:1008\n"
                     "ds_read_b128 %1, %8 offset:288\n ds_read_b96 %7, %8 offset:432\n ds_read_b128 %2, %8 offset:576\n ds_read_b128 %3, %8 offset:720\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                     : "v"(off * 4)
                     : "memory");
        gsum += gv;
        for (int k = 0; k < 4; ++k) {
            bad += a0[k] != pat(off + k);
            bad += a1[k] != pat(off + 72 + k);
            bad += a2[k] != pat(off + 144 + k);
            bad += a3[k] != pat(off + 180 + k);
        }
        for (int k = 0; k < 3; ++k) {
            bad += b0[k] != pat(off + 36 + k);
            bad += b1[k] != pat(off + 216 + k);
            bad += b2[k] != pat(off + 252 + k);
            bad += b3[k] != pat(off + 108 + k);
        }
    }
    if (gsum == 0x12345u) bad_by_lane[63] = 1;
    if (bad) atomicAdd(&bad_by_lane[lane], (unsigned long long)bad);
}

template <int SHAPE> static void run_victim(const char *name, int dmode, const uint32_t *src, size_t nbytes, uint32_t *sink,
                                            unsigned long long *d_bad, hipStream_t sa, hipStream_t sb)
{
    CK(hipMemset(d_bad, 0, 64 * sizeof(unsigned long long)));
    CK(hipDeviceSynchronize());
    const int dgrid = 4096, diters = 200;
    if (dmode == 0) hipLaunchKernelGGL(disturber<0>, dim3(dgrid), dim3(256), 32768, sa, src, nbytes, diters, sink);
    else if (dmode == 1) hipLaunchKernelGGL(disturber<1>, dim3(dgrid), dim3(256), 32768, sa, src, nbytes, diters, sink);
    else if (dmode == 2) hipLaunchKernelGGL(disturber<2>, dim3(dgrid), dim3(256), 32768, sa, src, nbytes, diters, sink);
    else if (dmode == 4) hipLaunchKernelGGL(disturber<3>, dim3(dgrid), dim3(256), 32768, sa, src, nbytes, diters, sink);
    else if (dmode == 5) hipLaunchKernelGGL(disturber<4>, dim3(dgrid), dim3(256), 32768, sa, src, nbytes, diters, sink);
    else if (dmode == 6) hipLaunchKernelGGL(disturber<5>, dim3(dgrid), dim3(256), 32768, sa, src, nbytes, diters, sink);
    else if (dmode == 7) hipLaunchKernelGGL(disturber<6>, dim3(dgrid), dim3(256), 32768, sa, src, nbytes, diters, sink);
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(victim<SHAPE>, dim3(4096), dim3(256), 0, sb, 2000, d_bad);
    CK(hipDeviceSynchronize());
    unsigned long long h[64];
    CK(hipMemcpy(h, d_bad, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
    printf("  %-44s bad reads %10llu   by lane quarter [%llu %llu %llu %llu]\n", name, tot, q[0], q[1], q[2], q[3]);
}


// ---- the kernel in which the problem was first seen (the MobileNet first conv with its weights as floats in LDS),
//      as a victim: run against a reference computed without a disturber ----
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
typedef __attribute__((ext_vector_type(4))) __bf16 hz_bf16x4;
__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d)
{
    hz_bf16x4 v = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
    return __builtin_bit_cast(uint2, v);
}
template <int VARIANT>
__global__ void __launch_bounds__(256) victim_f3(const uint16_t *__restrict__ x4, const uint16_t *__restrict__ w, // [8][3][3][4]
                                                       const float *__restrict__ bias, uint16_t *__restrict__ y, int B,
                                                       int H, int W, int Ho, int Wo, int Cd)
{
    // VARIANT 0: as shipped then (1152 B of LDS, compiler-chosen ds_read_b128/b96); 1: LDS allocation padded to 8 KiB;
    // 2: table placed 4 KiB into the allocation
    // round-2 variants (all with explicit broadcast ds_read_b128 unless noted):
    //  8: every wave sleeps ~25 us between the table fill + barrier and its first read ("errors sit early in a wave's
    //     life, around the fill + barrier" would make this one clean);  9: 32 rounds like 7, but the FIRST round's result
    //     is kept (early-life reads of a long-lived wave);  10: no divergent branch around the reads -- every lane
    //     executes all nine taps with EXEC all ones, out-of-image taps are zeroed arithmetically;  11: one read in
    //     flight at a time (each ds_read_b128 followed by its own lgkmcnt(0));  12: as 3 with the LDS allocation padded
    //     to 8 KiB (allocation granule / neighbouring workgroup's LDS)
    __shared__ __attribute__((aligned(16))) float ws_all[(VARIANT == 1 || VARIANT == 2 || VARIANT == 12) ? 2048 : 8 * 36];
    float *ws = ws_all + (VARIANT == 2 ? 1024 : 0);
    if (VARIANT == 1 || VARIANT == 12) ws_all[2047 - (threadIdx.x & 255)] = 0.f;
    for (int i = threadIdx.x; i < 8 * 36; i += 256) ws[i] = bf16_bits_to_f32(w[i]);
    __syncthreads();
    if (VARIANT == 8) {
        for (int k = 0; k < 400; ++k) __builtin_amdgcn_s_sleep(127); // 400 x 127 x 64 clocks ~ 25 us at 2.1 GHz
    }
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * Ho * Wo) return;
    const int wo = (int)(i % Wo);
    const int ho = (int)((i / Wo) % Ho);
    const int b = (int)(i / ((long long)Wo * Ho));
    float acc[8], first[8];
    for (int life = 0; life < ((VARIANT == 7 || VARIANT == 9) ? 32 : 1); ++life) { // 7 / 9: long-lived waves; 7 keeps the last round, 9 the first
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = bias[c];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hi = 2 * ho - 1 + ky;
        if (VARIANT != 10 && (unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int wi = 2 * wo - 1 + kx;
            if (VARIANT != 10 && (unsigned)wi >= (unsigned)W) continue;
            uint2 p;
            if (VARIANT == 10) { // EXEC stays all ones: clamped address, pixel zeroed when the tap is outside the image
                const bool in = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
                p = reinterpret_cast<const uint2 *>(x4)[((long long)b * H + (in ? hi : 0)) * W + (in ? wi : 0)];
                if (!in) p = make_uint2(0, 0);
            } else if (VARIANT == 6) { // no global load in the loop: a pixel made from the coordinates (bf16 2.0 .. 3.98)
                p.x = (0x4000u + (uint32_t)((hi * 7 + wi * 3) & 0x7f)) | ((0x4000u + (uint32_t)((hi + wi) & 0x7f)) << 16);
                p.y = 0x4000u + (uint32_t)((hi * 5 + wi) & 0x7f);
            } else {
                p = reinterpret_cast<const uint2 *>(x4)[((long long)b * H + hi) * W + wi];
            }
            const float r = bf16_bits_to_f32(p.x & 0xffffu), g = bf16_bits_to_f32(p.x >> 16), bl = bf16_bits_to_f32(p.y & 0xffffu);
            if (VARIANT >= 3) {
                // 3: the tap's 8 weight vectors by explicit broadcast ds_read_b128, ALL landed (lgkmcnt(0)) before any
                //    arithmetic; 4: the same with ds_read_b64 pairs; 5: b128, but the arithmetic of channel c starts
                //    while the reads of the later channels are still in flight (counted waits)
                typedef __attribute__((ext_vector_type(4))) float f4;
                typedef __attribute__((ext_vector_type(2))) float f2;
                const uint32_t a0 = (uint32_t)(uintptr_t)ws + (uint32_t)(ky * 3 + kx) * 16u;
                f4 wv[8];
                if (VARIANT == 4) {
                    f2 l0, l1, l2, l3, l4, l5, l6, l7, h0, h1, h2, h3, h4, h5, h6, h7;
                    asm volatile("ds_read_b64 %0, %16\n ds_read_b64 %8, %16 offset:8\n ds_read_b64 %1, %16 offset:144\n ds_read_b64 %9, %16 offset:152\n"
                                 "ds_read_b64 %2, %16 offset:288\n ds_read_b64 %10, %16 offset:296\n ds_read_b64 %3, %16 offset:432\n ds_read_b64 %11, %16 offset:440\n"
                                 "ds_read_b64 %4, %16 offset:576\n ds_read_b64 %12, %16 offset:584\n ds_read_b64 %5, %16 offset:720\n ds_read_b64 %13, %16 offset:728\n"
                                 "ds_read_b64 %6, %16 offset:864\n ds_read_b64 %14, %16 offset:872\n ds_read_b64 %7, %16 offset:1008\n ds_read_b64 %15, %16 offset:1016\n"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(l4), "=&v"(l5), "=&v"(l6), "=&v"(l7),
                                   "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(h4), "=&v"(h5), "=&v"(h6), "=&v"(h7)
                                 : "v"(a0) : "memory");
                    wv[0] = f4{l0.x, l0.y, h0.x, h0.y}; wv[1] = f4{l1.x, l1.y, h1.x, h1.y}; wv[2] = f4{l2.x, l2.y, h2.x, h2.y};
                    wv[3] = f4{l3.x, l3.y, h3.x, h3.y}; wv[4] = f4{l4.x, l4.y, h4.x, h4.y}; wv[5] = f4{l5.x, l5.y, h5.x, h5.y};
                    wv[6] = f4{l6.x, l6.y, h6.x, h6.y}; wv[7] = f4{l7.x, l7.y, h7.x, h7.y};
                } else if (VARIANT == 11) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=&v"(wv[c]) : "v"(a0 + (uint32_t)c * 144u) : "memory");
                } else if (VARIANT == 13) { // as 3, with ~64 idle cycles between the branch's EXEC write and the first read
                    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
                                 "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:144\n ds_read_b128 %2, %8 offset:288\n ds_read_b128 %3, %8 offset:432\n"
                                 "ds_read_b128 %4, %8 offset:576\n ds_read_b128 %5, %8 offset:720\n ds_read_b128 %6, %8 offset:864\n ds_read_b128 %7, %8 offset:1008\n"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(wv[0]), "=&v"(wv[1]), "=&v"(wv[2]), "=&v"(wv[3]), "=&v"(wv[4]), "=&v"(wv[5]), "=&v"(wv[6]), "=&v"(wv[7])
                                 : "v"(a0) : "memory");
                } else if (VARIANT == 14) { // as 3, with ~64 idle cycles between the last read's return and the EXEC restore
                    asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:144\n ds_read_b128 %2, %8 offset:288\n ds_read_b128 %3, %8 offset:432\n"
                                 "ds_read_b128 %4, %8 offset:576\n ds_read_b128 %5, %8 offset:720\n ds_read_b128 %6, %8 offset:864\n ds_read_b128 %7, %8 offset:1008\n"
                                 "s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15"
                                 : "=&v"(wv[0]), "=&v"(wv[1]), "=&v"(wv[2]), "=&v"(wv[3]), "=&v"(wv[4]), "=&v"(wv[5]), "=&v"(wv[6]), "=&v"(wv[7])
                                 : "v"(a0) : "memory");
                } else if (VARIANT == 3 || VARIANT == 6 || VARIANT == 7 || VARIANT == 8 || VARIANT == 9 || VARIANT == 10 || VARIANT == 12) {
                    asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:144\n ds_read_b128 %2, %8 offset:288\n ds_read_b128 %3, %8 offset:432\n"
                                 "ds_read_b128 %4, %8 offset:576\n ds_read_b128 %5, %8 offset:720\n ds_read_b128 %6, %8 offset:864\n ds_read_b128 %7, %8 offset:1008\n"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(wv[0]), "=&v"(wv[1]), "=&v"(wv[2]), "=&v"(wv[3]), "=&v"(wv[4]), "=&v"(wv[5]), "=&v"(wv[6]), "=&v"(wv[7])
                                 : "v"(a0) : "memory");
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(wv[c]) : "v"(a0 + (uint32_t)c * 144u) : "memory");
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (VARIANT == 5) {
                        // wait until read c has landed (7 - c younger reads may still be in flight)
                        switch (c) {
                        case 0: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(wv[0]) :: "memory"); break;
                        case 1: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wv[1]) :: "memory"); break;
                        case 2: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(wv[2]) :: "memory"); break;
                        case 3: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wv[3]) :: "memory"); break;
                        case 4: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wv[4]) :: "memory"); break;
                        case 5: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wv[5]) :: "memory"); break;
                        case 6: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wv[6]) :: "memory"); break;
                        default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wv[7]) :: "memory"); break;
                        }
                    }
                    acc[c] += r * wv[c].x + g * wv[c].y + bl * wv[c].z;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float *wc = ws + c * 36 + (ky * 3 + kx) * 4;
                    acc[c] += r * wc[0] + g * wc[1] + bl * wc[2];
                }
            }
        }
    }
    if (VARIANT == 7 || VARIANT == 9) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
    if (VARIANT == 9 && life == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) first[c] = acc[c];
    }
    }
    if (VARIANT == 9) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = first[c];
    }
    uint4 *dst = reinterpret_cast<uint4 *>(y + i * Cd);
    const uint2 lo = pack_bf16x4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    const uint2 hi2 = pack_bf16x4(fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f));
    dst[0] = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
    for (int k = 1; k < Cd / 8; ++k) dst[k] = make_uint4(0, 0, 0, 0);
}

__global__ void hz_compare(const uint16_t *a, const uint16_t *b, size_t n_px, int Cd, unsigned long long *bad_by_lane)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; // pixel
    if (i >= n_px) return;
    unsigned bad = 0;
    for (int c = 0; c < 8; ++c) bad += a[i * Cd + c] != b[i * Cd + c];
    if (bad) atomicAdd(&bad_by_lane[i & 63], (unsigned long long)bad);
}

struct F3State { uint16_t *x4, *w, *y, *yref; float *bias; unsigned long long *bad; hipStream_t s; int B, H, W, Ho, Wo, Cd; };
static F3State g_f3;

template <int V> static void f3_launch(uint16_t *y)
{
    const long long total = (long long)g_f3.B * g_f3.Ho * g_f3.Wo;
    hipLaunchKernelGGL(victim_f3<V>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, g_f3.s, g_f3.x4, g_f3.w, g_f3.bias, y,
                       g_f3.B, g_f3.H, g_f3.W, g_f3.Ho, g_f3.Wo, g_f3.Cd);
}

// reference run: call BEFORE any disturber is started
extern "C" __attribute__((visibility("default"))) int hazard_f3_init(void)
{
    F3State &f = g_f3;
    f.B = 8; f.H = 640; f.W = 640; f.Ho = 320; f.Wo = 320; f.Cd = 64;
    const size_t nx = (size_t)f.B * f.H * f.W * 4, ny = (size_t)f.B * f.Ho * f.Wo * f.Cd;
    CK(hipStreamCreateWithFlags(&f.s, hipStreamNonBlocking));
    CK(hipMalloc(&f.x4, nx * 2)); CK(hipMalloc(&f.w, 288 * 2)); CK(hipMalloc(&f.bias, 8 * 4));
    CK(hipMalloc(&f.y, ny * 2)); CK(hipMalloc(&f.yref, ny * 2)); CK(hipMalloc(&f.bad, 64 * 8));
    std::vector<uint16_t> hx(nx), hw(288);
    uint32_t r = 12345u;
    for (auto &v : hx) { r = r * 1664525u + 1013904223u; v = (uint16_t)(0x4000u + ((r >> 20) & 0x3ffu)); } // bf16 in [2, 4)
    for (auto &v : hw) { r = r * 1664525u + 1013904223u; v = (uint16_t)(0x3c00u + ((r >> 20) & 0x3ffu) + ((r >> 31) << 15)); }
    float hb[8] = {0.1f, -0.2f, 0.3f, 0.05f, -0.4f, 0.2f, 0.0f, 0.15f};
    CK(hipMemcpy(f.x4, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(f.w, hw.data(), 288 * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(f.bias, hb, sizeof hb, hipMemcpyHostToDevice));
    f3_launch<0>(f.yref);
    CK(hipStreamSynchronize(f.s));
    return 0;
}

// reference for a variant whose arithmetic differs from variant 0 (6: synthetic pixels): run it once UNDISTURBED
extern "C" __attribute__((visibility("default"))) int hazard_f3_reference(int variant)
{
    F3State &f = g_f3;
    if (variant == 6) f3_launch<6>(f.yref); else f3_launch<0>(f.yref);
    CK(hipStreamSynchronize(f.s));
    return 0;
}

extern "C" __attribute__((visibility("default"))) int hazard_f3_run(int variant, int launches, unsigned long long *out64)
{
    F3State &f = g_f3;
    CK(hipMemsetAsync(f.bad, 0, 64 * 8, f.s));
    const size_t npx = (size_t)f.B * f.Ho * f.Wo;
    for (int i = 0; i < launches; ++i) {
        if (variant == 0) f3_launch<0>(f.y);
        else if (variant == 1) f3_launch<1>(f.y);
        else if (variant == 2) f3_launch<2>(f.y);
        else if (variant == 3) f3_launch<3>(f.y);
        else if (variant == 4) f3_launch<4>(f.y);
        else if (variant == 5) f3_launch<5>(f.y);
        else if (variant == 6) f3_launch<6>(f.y);
        else if (variant == 8) f3_launch<8>(f.y);
        else if (variant == 9) f3_launch<9>(f.y);
        else if (variant == 10) f3_launch<10>(f.y);
        else if (variant == 11) f3_launch<11>(f.y);
        else if (variant == 12) f3_launch<12>(f.y);
        else if (variant == 13) f3_launch<13>(f.y);
        else if (variant == 14) f3_launch<14>(f.y);
        else f3_launch<7>(f.y);
        hipLaunchKernelGGL(hz_compare, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, f.s, f.y, f.yref, npx, f.Cd, f.bad);
    }
    CK(hipMemcpyAsync(out64, f.bad, 64 * 8, hipMemcpyDeviceToHost, f.s));
    CK(hipStreamSynchronize(f.s));
    return 0;
}

// ---- shared-library entry (tools/hazard_with_conv.py): run one victim shape on its own stream while the caller
//      keeps real conv kernels of librfd_hip.so in flight from another thread ----
extern "C" __attribute__((visibility("default"))) int hazard_victim(int shape, int launches, int iters, unsigned long long *out64)
{
    static unsigned long long *d_bad = nullptr;
    static hipStream_t sb = nullptr;
    static uint32_t *d_g = nullptr;
    if (!d_g) { CK(hipMalloc(&d_g, 4u << 20)); CK(hipMemset(d_g, 1, 4u << 20)); }
    if (!d_bad) { CK(hipMalloc(&d_bad, 64 * sizeof(unsigned long long))); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
    CK(hipMemsetAsync(d_bad, 0, 64 * sizeof(unsigned long long), sb));
    for (int rep = 0; rep < launches; ++rep) {
        switch (shape) {
        case 0: hipLaunchKernelGGL(victim<0>, dim3(4096), dim3(256), 0, sb, iters, d_bad); break;
        case 1: hipLaunchKernelGGL(victim<1>, dim3(4096), dim3(256), 0, sb, iters, d_bad); break;
        case 2: hipLaunchKernelGGL(victim<2>, dim3(4096), dim3(256), 0, sb, iters, d_bad); break;
        case 3: hipLaunchKernelGGL(victim<3>, dim3(4096), dim3(256), 0, sb, iters, d_bad); break;
        case 4: hipLaunchKernelGGL(victim<4>, dim3(4096), dim3(256), 0, sb, iters, d_bad); break;
        case 5: hipLaunchKernelGGL(victim<5>, dim3(4096), dim3(256), 0, sb, iters, d_bad); break;
        case 6: hipLaunchKernelGGL(victim<6>, dim3(4096), dim3(256), 0, sb, iters, d_bad); break;
        case 7: hipLaunchKernelGGL(victim<7>, dim3(4096), dim3(256), 0, sb, iters, d_bad); break;
        case 8: hipLaunchKernelGGL(victim_multi<0>, dim3(4096), dim3(256), 0, sb, iters, d_bad, (const uint32_t *)d_g); break;
        default: hipLaunchKernelGGL(victim_multi<1>, dim3(4096), dim3(256), 0, sb, iters, d_bad, (const uint32_t *)d_g); break;
        }
    }
    CK(hipMemcpyAsync(out64, d_bad, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, sb));
    CK(hipStreamSynchronize(sb));
    return 0;
}

// ---- does the consumer matter?  broadcast ds_read_b128 of FLOATS, all landed, then the loaded registers go through
//      v_pk_mul_f32 (PK = 1: packed, on the 64-bit halves of the loaded tuple), v_mul_f32 (PK = 0) or straight into a
//      compare (PK = 2); 8 reads per step at 144-byte records like the first-conv kernel ----
template <int PK> __global__ void __launch_bounds__(256) victim_pk(int iters, unsigned long long *bad_by_lane)
{
    __shared__ __attribute__((aligned(16))) float tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = 1.0f + 0.5f * (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    typedef __attribute__((ext_vector_type(4))) float f4;
    typedef __attribute__((ext_vector_type(2))) float f2;
    unsigned bad = 0;
    const f2 two = {2.0f, 2.0f};
    for (int it = 0; it < iters; ++it) {
        const int off = (it & 3) * 4; // dwords: tap 0..3 of eight 36-dword records
        f4 w[8];
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:144\n ds_read_b128 %2, %8 offset:288\n ds_read_b128 %3, %8 offset:432\n"
                     "ds_read_b128 %4, %8 offset:576\n ds_read_b128 %5, %8 offset:720\n ds_read_b128 %6, %8 offset:864\n ds_read_b128 %7, %8 offset:1008\n"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                     : "v"((uint32_t)(uintptr_t)tab + (uint32_t)off * 4u) : "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int d = off + c * 36;
            float r[4];
            if (PK == 1) {
                f2 lo = {w[c].x, w[c].y}, hi = {w[c].z, w[c].w}, plo, phi;
                asm volatile("v_pk_mul_f32 %0, %2, %4\n v_pk_mul_f32 %1, %3, %4" : "=&v"(plo), "=&v"(phi) : "v"(lo), "v"(hi), "v"(two));
                r[0] = plo.x; r[1] = plo.y; r[2] = phi.x; r[3] = phi.y;
            } else if (PK == 0) {
                asm volatile("v_mul_f32 %0, 2.0, %4\n v_mul_f32 %1, 2.0, %5\n v_mul_f32 %2, 2.0, %6\n v_mul_f32 %3, 2.0, %7"
                             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(w[c].x), "v"(w[c].y), "v"(w[c].z), "v"(w[c].w));
            } else {
                r[0] = 2.0f * w[c].x; r[1] = 2.0f * w[c].y; r[2] = 2.0f * w[c].z; r[3] = 2.0f * w[c].w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) bad += r[k] != 2.0f * (1.0f + 0.5f * (float)(d + k));
        }
    }
    if (bad) atomicAdd(&bad_by_lane[lane], (unsigned long long)bad);
}

static void launch_disturber(int dmode, int grid, int iters, const uint32_t *src, size_t nbytes, uint32_t *sink, hipStream_t sa)
{
    switch (dmode) {
    case 0: hipLaunchKernelGGL(disturber<0>, dim3(grid), dim3(256), 32768, sa, src, nbytes, iters, sink); break;
    case 1: hipLaunchKernelGGL(disturber<1>, dim3(grid), dim3(256), 32768, sa, src, nbytes, iters, sink); break;
    case 2: hipLaunchKernelGGL(disturber<2>, dim3(grid), dim3(256), 32768, sa, src, nbytes, iters, sink); break;
    case 4: hipLaunchKernelGGL(disturber<3>, dim3(grid), dim3(256), 32768, sa, src, nbytes, iters, sink); break;
    case 5: hipLaunchKernelGGL(disturber<4>, dim3(grid), dim3(256), 32768, sa, src, nbytes, iters, sink); break;
    case 6: hipLaunchKernelGGL(disturber<5>, dim3(grid), dim3(256), 32768, sa, src, nbytes, iters, sink); break;
    case 7: hipLaunchKernelGGL(disturber<6>, dim3(grid), dim3(256), 32768, sa, src, nbytes, iters, sink); break;
    default: break;
    }
}

static void run_multi(int dmode, const uint32_t *src, size_t nbytes, uint32_t *sink, unsigned long long *d_bad, hipStream_t sa, hipStream_t sb)
{
    static uint32_t *d_g = nullptr;
    if (!d_g) { CK(hipMalloc(&d_g, 4u << 20)); CK(hipMemset(d_g, 1, 4u << 20)); }
    for (int with_vmem = 0; with_vmem < 2; ++with_vmem) {
        CK(hipMemset(d_bad, 0, 64 * sizeof(unsigned long long)));
        CK(hipDeviceSynchronize());
        launch_disturber(dmode, 4096, 200, src, nbytes, sink, sa);
        for (int rep = 0; rep < 4; ++rep) {
            if (with_vmem) hipLaunchKernelGGL(victim_multi<1>, dim3(4096), dim3(256), 0, sb, 2000, d_bad, (const uint32_t *)d_g);
            else hipLaunchKernelGGL(victim_multi<0>, dim3(4096), dim3(256), 0, sb, 2000, d_bad, (const uint32_t *)d_g);
        }
        CK(hipDeviceSynchronize());
        unsigned long long h[64], tot = 0, q[4] = {0, 0, 0, 0};
        CK(hipMemcpy(h, d_bad, sizeof h, hipMemcpyDeviceToHost));
        for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
        printf("  %-44s bad reads %10llu   by lane quarter [%llu %llu %llu %llu]\n",
               with_vmem ? "8 broadcast reads in flight + a global load" : "8 broadcast reads in flight (4 b128 + 4 b96)", tot, q[0], q[1], q[2], q[3]);
    }
}

// round 2: only the first-conv victims, undisturbed and under the two MFMA disturbers, with the discriminating variants
static int argc_quick2 = 0; // "quick2": skip the variants already settled
static int quick_main(const uint32_t *src, size_t nbytes, uint32_t *sink, hipStream_t sa)
{
    if (!argc_quick2) { // the minimal form first: masked lanes around a plain read-and-compare loop, MFMA-on-registers disturber
        unsigned long long *d_bad;
        hipStream_t sb;
        CK(hipMalloc(&d_bad, 64 * sizeof(unsigned long long)));
        CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        struct Case { const char *name; int width, addr; unsigned long long mask; };
        const Case cases[] = {
            {"broadcast b128, EXEC all ones", 128, 0, 0ull},
            {"broadcast b128, lane 0 off", 128, 0, 1ull},
            {"broadcast b128, lane 63 off", 128, 0, 1ull << 63},
            {"broadcast b128, lane 20 off", 128, 0, 1ull << 20},
            {"broadcast b128, lanes 0-31 off", 128, 0, 0xffffffffull},
            {"broadcast b128, lanes 48-63 off", 128, 0, 0xffffull << 48},
            {"per-lane b128 (lane*16), lane 0 off", 128, 1, 1ull},
            {"fragment b128 ((l&15)*64+(l>>4)*16), lane 0 off", 128, 2, 1ull},
            {"fragment b128, lanes 0-15 off", 128, 2, 0xffffull},
            {"broadcast 2 x b64, lane 0 off", 64, 0, 1ull},
            {"fragment 2 x b64, lane 0 off", 64, 2, 1ull},
        };
        for (int d = 0; d < 2; ++d) {
            printf("minimal form, disturber: %s\n", d ? "32 MFMA per step on register operands only (no LDS traffic)" : "none");
            for (const Case &c : cases) {
                CK(hipMemset(d_bad, 0, 64 * sizeof(unsigned long long)));
                CK(hipDeviceSynchronize());
                if (d) hipLaunchKernelGGL(disturber<6>, dim3(4096), dim3(256), 32768, sa, src, nbytes, 200, sink);
                for (int rep = 0; rep < 4; ++rep) {
                    if (c.width == 128 && c.addr == 0) hipLaunchKernelGGL((victim_masked<128, 0>), dim3(4096), dim3(256), 0, sb, 2000, c.mask, d_bad);
                    else if (c.width == 128 && c.addr == 1) hipLaunchKernelGGL((victim_masked<128, 1>), dim3(4096), dim3(256), 0, sb, 2000, c.mask, d_bad);
                    else if (c.width == 128) hipLaunchKernelGGL((victim_masked<128, 2>), dim3(4096), dim3(256), 0, sb, 2000, c.mask, d_bad);
                    else if (c.addr == 0) hipLaunchKernelGGL((victim_masked<64, 0>), dim3(4096), dim3(256), 0, sb, 2000, c.mask, d_bad);
                    else hipLaunchKernelGGL((victim_masked<64, 2>), dim3(4096), dim3(256), 0, sb, 2000, c.mask, d_bad);
                }
                CK(hipDeviceSynchronize());
                unsigned long long h[64], tot = 0, q[4] = {0, 0, 0, 0};
                CK(hipMemcpy(h, d_bad, sizeof h, hipMemcpyDeviceToHost));
                for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
                printf("  %-52s bad reads %10llu  by lane quarter [%llu %llu %llu %llu]\n", c.name, tot, q[0], q[1], q[2], q[3]);
                fflush(stdout);
            }
        }
    }
    hazard_f3_init();
    const char *fn[12] = {"explicit broadcast b128 (as round 1)", "explicit 2 x b64", "long-lived (32 rounds), LAST round kept",
                          "long-lived (32 rounds), FIRST round kept", "25 us sleep between fill+barrier and the first read",
                          "EXEC all ones (no branch around the reads)", "one b128 in flight at a time", "LDS allocation padded to 8 KiB",
                          "compiler-chosen reads", "b128, counted waits (reads still in flight)",
                          "64 idle cycles between the EXEC write and the reads", "64 idle cycles after the reads landed"};
    const int fv[12] = {3, 4, 7, 9, 8, 10, 11, 12, 0, 5, 13, 14};
    const int dmodes[3] = {3, 7, 5};
    const char *dn[3] = {"none", "32 MFMA per step on register operands only (no LDS traffic)", "LDS-DMA dwordx4 + fragment reads + 32 MFMA per step"};
    for (int d = 0; d < 3; ++d) {
        printf("disturber: %s\n", dn[d]);
        for (int k = 0; k < 12; ++k) {
            if (argc_quick2 && k != 0 && k != 5 && k < 10) continue;
            CK(hipDeviceSynchronize());
            hazard_f3_reference(fv[k]);
            if (dmodes[d] == 7) hipLaunchKernelGGL(disturber<6>, dim3(8192), dim3(256), 32768, sa, src, nbytes, fv[k] == 8 ? 1600 : 400, sink);
            else if (dmodes[d] == 5) hipLaunchKernelGGL(disturber<4>, dim3(8192), dim3(256), 32768, sa, src, nbytes, fv[k] == 8 ? 1600 : 400, sink);
            unsigned long long h[64];
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, g_f3.s));
            hazard_f3_run(fv[k], 12, h);
            CK(hipEventRecord(e1, g_f3.s));
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
            printf("  %-52s bad elements %8llu  by lane quarter [%llu %llu %llu %llu]  (12 launches, %.1f ms)\n", fn[k], tot, q[0], q[1], q[2], q[3], ms);
            fflush(stdout);
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    const size_t nbytes = 256u << 20;
    uint32_t *src, *sink;
    unsigned long long *d_bad;
    CK(hipMalloc(&src, nbytes));
    CK(hipMemset(src, 0x5a, nbytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&d_bad, 64 * sizeof(unsigned long long)));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    if (argc > 1 && !strcmp(argv[1], "quick2")) argc_quick2 = 1;
    if (argc > 1 && !strncmp(argv[1], "quick", 5)) return quick_main(src, nbytes, sink, sa);
    const char *dn[8] = {"LDS-DMA dwordx4 (buffer_load_dwordx4 lds)", "LDS-DMA dword (buffer_load_dword lds)", "global_load + ds_write_b128", "none",
                         "LDS-DMA dwordx4 + 16 fragment ds_read_b128 per step", "LDS-DMA dwordx4 + fragment reads + 32 MFMA per step",
                         "fragment ds_read_b128 + 32 MFMA per step, no DMA", "32 MFMA per step on register operands only (no LDS traffic)"};
    for (int dmode = 0; dmode < 8; ++dmode) {
        printf("disturber: %s\n", dn[dmode]);
        run_victim<0>("broadcast ds_read_b128", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<7>("two-address ds_read_b128 (lane>>5)", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<1>("per-lane ds_read_b128 (lane*16)", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<5>("fragment ds_read_b128 ((l&15)*64+(l>>4)*16)", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<4>("broadcast ds_read_b96", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<2>("broadcast ds_read_b64", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<6>("per-lane ds_read_b64", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<3>("broadcast ds_read_b32", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_multi(dmode, src, nbytes, sink, d_bad, sa, sb);
        for (int pk = 0; pk < 3; ++pk) {
            CK(hipMemset(d_bad, 0, 64 * sizeof(unsigned long long)));
            CK(hipDeviceSynchronize());
            launch_disturber(dmode, 4096, 200, src, nbytes, sink, sa);
            for (int rep = 0; rep < 4; ++rep) {
                if (pk == 0) hipLaunchKernelGGL(victim_pk<0>, dim3(4096), dim3(256), 0, sb, 500, d_bad);
                else if (pk == 1) hipLaunchKernelGGL(victim_pk<1>, dim3(4096), dim3(256), 0, sb, 500, d_bad);
                else hipLaunchKernelGGL(victim_pk<2>, dim3(4096), dim3(256), 0, sb, 500, d_bad);
            }
            CK(hipDeviceSynchronize());
            unsigned long long h[64], tot = 0, q[4] = {0, 0, 0, 0};
            CK(hipMemcpy(h, d_bad, sizeof h, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
            static const char *pn[3] = {"8 broadcast b128 of floats -> v_mul_f32", "8 broadcast b128 of floats -> v_pk_mul_f32", "8 broadcast b128 of floats -> compiler math"};
            printf("  %-44s bad values %9llu   by lane quarter [%llu %llu %llu %llu]\n", pn[pk], tot, q[0], q[1], q[2], q[3]);
        }
        // the first-conv kernel variants (hazard_f3_*), same synthetic disturbers
        if (dmode == 0) hazard_f3_init();
        const char *fn[5] = {"first-conv kernel, compiler-chosen reads", "first-conv kernel, explicit broadcast b128", "first-conv kernel, explicit 2 x b64",
                             "explicit b128, no global loads in the loop", "explicit b128, long-lived waves (32 rounds)"};
        const int fv[5] = {0, 3, 4, 6, 7};
        for (int k = 0; k < 5; ++k) {
            CK(hipDeviceSynchronize());
            hazard_f3_reference(fv[k]);
            if (dmode == 0) hipLaunchKernelGGL(disturber<0>, dim3(8192), dim3(256), 32768, sa, src, nbytes, 400, sink);
            else if (dmode == 1) hipLaunchKernelGGL(disturber<1>, dim3(8192), dim3(256), 32768, sa, src, nbytes, 400, sink);
            else if (dmode == 2) hipLaunchKernelGGL(disturber<2>, dim3(8192), dim3(256), 32768, sa, src, nbytes, 400, sink);
            else if (dmode == 4) hipLaunchKernelGGL(disturber<3>, dim3(8192), dim3(256), 32768, sa, src, nbytes, 400, sink);
            else if (dmode == 5) hipLaunchKernelGGL(disturber<4>, dim3(8192), dim3(256), 32768, sa, src, nbytes, 400, sink);
            else if (dmode == 6) hipLaunchKernelGGL(disturber<5>, dim3(8192), dim3(256), 32768, sa, src, nbytes, 400, sink);
            else if (dmode == 7) hipLaunchKernelGGL(disturber<6>, dim3(8192), dim3(256), 32768, sa, src, nbytes, 400, sink);
            unsigned long long h[64];
            hazard_f3_run(fv[k], 12, h);
            CK(hipDeviceSynchronize());
            unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
            printf("  %-44s bad elements %8llu   by lane quarter [%llu %llu %llu %llu]\n", fn[k], tot, q[0], q[1], q[2], q[3]);
        }
    }
    return 0;
}

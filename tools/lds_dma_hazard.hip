// lds_dma_hazard.hip -- micro-test written while chasing a cross-workgroup problem on gfx950 (MI355X, ROCm 7.0
// runtime): the old MobileNet first-conv kernel kept its 288 weights as floats in LDS and read them with broadcast
// ds_read_b128; whenever a conv workgroup of ANOTHER stream (tiles streamed in with LDS-DMA, buffer_load_dwordx4 ...
// lds) shared the CU, ~1e-4 of those wave reads returned wrong data, always in lanes 48..63 and only for the b128
// reads (the ds_read_b96 reads of the same table, and a build without ds_read_b128, were clean).
//
// RESULT OF THIS PROGRAM: it does NOT reproduce the problem -- every read shape below comes back clean, with the
// synthetic disturbers here and with the real conv kernels of librfd_hip.so in flight (tools/_hazard_with_conv.py).
// The trigger therefore needs something of the real kernel that is not modelled here; the product avoids it by
// keeping wave-uniform tables out of LDS (scalar loads) and tests/test_concurrency_gpu.py guards the pipeline.
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_hazard tools/lds_dma_hazard.hip && ./lds_dma_hazard
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/liblds_dma_hazard.so tools/lds_dma_hazard.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- disturbers ----
// mode 0: LDS-DMA dwordx4; mode 1: LDS-DMA dword; mode 2: global_load + ds_write_b128 (no DMA)
template <int MODE> __global__ void __launch_bounds__(256) disturber(const uint32_t *src, size_t n_bytes, int iters, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, (uint32_t)n_bytes, 0x00020000);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t base = (uint32_t)(((size_t)(blockIdx.x * 131 + it) * 4096) % (n_bytes - 65536));
        for (int q = 0; q < 8; ++q) {
            unsigned char *dst = smem + (wave * 8 + q) * 1024;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)dst, 16,
                                                         base + (wave * 8 + q) * 1024 + lane * 16, 0, 0, 0);
            } else if (MODE == 1) {
                for (int k = 0; k < 4; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(dst + k * 256), 4,
                                                             base + (wave * 8 + q) * 1024 + k * 256 + lane * 4, 0, 0, 0);
            } else {
                const uint4 v = *reinterpret_cast<const uint4 *>((const char *)src + base + (wave * 8 + q) * 1024 + lane * 16);
                *reinterpret_cast<uint4 *>(dst + lane * 16) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<const uint32_t *>(smem + ((threadIdx.x * 52 + it * 4) & 32764));
        __syncthreads();
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
    for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(victim<SHAPE>, dim3(4096), dim3(256), 0, sb, 2000, d_bad);
    CK(hipDeviceSynchronize());
    unsigned long long h[64];
    CK(hipMemcpy(h, d_bad, sizeof h, hipMemcpyDeviceToHost));
    unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
    printf("  %-44s bad reads %10llu   by lane quarter [%llu %llu %llu %llu]\n", name, tot, q[0], q[1], q[2], q[3]);
}

// ---- shared-library entry (tools/_hazard_with_conv.py): run one victim shape on its own stream while the caller
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

int main()
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
    const char *dn[4] = {"LDS-DMA dwordx4 (buffer_load_dwordx4 lds)", "LDS-DMA dword (buffer_load_dword lds)", "global_load + ds_write_b128", "none"};
    for (int dmode = 0; dmode < 4; ++dmode) {
        printf("disturber: %s\n", dn[dmode]);
        run_victim<0>("broadcast ds_read_b128", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<7>("two-address ds_read_b128 (lane>>5)", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<1>("per-lane ds_read_b128 (lane*16)", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<5>("fragment ds_read_b128 ((l&15)*64+(l>>4)*16)", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<4>("broadcast ds_read_b96", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<2>("broadcast ds_read_b64", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<6>("per-lane ds_read_b64", dmode, src, nbytes, sink, d_bad, sa, sb);
        run_victim<3>("broadcast ds_read_b32", dmode, src, nbytes, sink, d_bad, sa, sb);
    }
    return 0;
}

// mfma_lds_loop.hip -- what the K-step body of the 256 x 256 persistent kernels (pw_wide_kernel; conv3x3_halo_kernel's is the same
// shape) costs IN ISOLATION: 8 waves = 4 (pixel) x 2 (channel), a wave owns 64 pixels x 128 channels = 32 accumulators; per step
// it reads 8 B fragments + 16 A fragments (ds_read_b128, the kernels' swizzled conflict-free layout) and issues 64
// v_mfma_f32_16x16x32_bf16 -- the software-pipelined order of the kernels (fragments of group g + 1 read while group g's 16 MFMAs
// issue).  No global loads, no LDS-DMA, operands random bf16 staged once; one 512-thread workgroup per CU.
// Variants: A = the body as shipped, waves free-running; B = + the step barrier (s_waitcnt lgkmcnt(0); s_barrier);
// C = one wave per SIMD (4 waves, twice the steps); D = no LDS reads in the loop (fragments read once): the MFMA issue alone;
// E = A with 32 KiB of LDS-DMA per step issued by nobody but with 16 extra ds_write_b128 per wave and step (the LDS write
// bandwidth a step's incoming operands take) -- not built: the DMA's own cost is measured in the kernels (RFD_WIDE_EXP).
// Prints cycles per step (median over workgroups, s_memtime), the clock held (s_memtime / s_memrealtime) and the PF/s implied.
// The ideal is 64 MFMAs x 16 cycles x 2 waves per SIMD = 2 048 cycles per step.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_lds_loop tools/mfma_lds_loop.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16_t;

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, uint32_t soffset, void *lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, soffset, 0, 0);
}

struct Args {
    const bf16_t *src;  // 128 KiB of random bf16 (variants E, F: the L2-resident source of every step's 64 KiB of operands)
    float *sink;        // [grid][512][4]
    unsigned long long *stamps; // [grid][2]: d s_memtime, d s_memrealtime
    int nsteps;
};

template <int VAR>
__global__ void __launch_bounds__(512) loop_kernel(const Args a)
{
    constexpr int XEL = 256 * 64, WEL = 256 * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem); // [2][XEL]
    bf16_t *Ws = Xs + 2 * XEL;                     // [2][WEL]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = (wave >> 2) & 1;
    const int frow = lane & 15, fq = lane >> 4;
    for (int i = tid; i < (2 * XEL + 2 * WEL) / 8; i += blockDim.x)
        reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(a.src)[i];
    __syncthreads();
    const int xb0 = (wm * 64 + frow) * 64 + ((fq ^ (frow & 7)) << 3), xb1 = xb0 ^ 32;
    const int wa0 = (wn * 128 + frow) * 64 + ((fq ^ (frow & 7)) << 3), wa1 = wa0 ^ 32;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int sl = 0;
    const int nsteps = VAR == 2 ? a.nsteps * 2 : a.nsteps;
    bf16x8 af[2][4], bfr[2][4];
    if (VAR == 3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { bfr[0][j] = *reinterpret_cast<const bf16x8 *>(Xs + j * 1024 + xb0); bfr[1][j] = *reinterpret_cast<const bf16x8 *>(Xs + j * 1024 + xb1); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { af[0][i] = *reinterpret_cast<const bf16x8 *>(Ws + i * 1024 + wa0); af[1][i] = *reinterpret_cast<const bf16x8 *>(Ws + (4 + i) * 1024 + wa1); }
    }
    __syncthreads();
    // variants E, F: the kernels' operand stream -- every wave requests 8 of the NEXT step's 64 1-KiB pieces (4 activation, 4 weight)
    // into the other slot pair behind the step barrier and drains them at the top of the next step
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.src), 0, 128 * 1024, 0x00020000);
    const uint32_t dlane = (uint32_t)(lane * 16);
    auto issue_piece = [&](int slot, int idx) __attribute__((always_inline)) {
        const int q = idx & 3;
        bf16_t *dst = (idx < 4 ? Xs + slot * XEL : Ws + slot * WEL) + (wave + 8 * q) * 512;
        blds16(rs, dlane, (uint32_t)(((idx < 4 ? 0 : 2 * XEL) + slot * XEL + (wave + 8 * q) * 512) * 2), dst);
    };
    constexpr bool DMA = VAR == 4 || VAR == 5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < nsteps; ++s) {
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (VAR == 1 || DMA) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (VAR == 4) {
#pragma unroll
            for (int idx = 0; idx < 8; ++idx) issue_piece(sl ^ 1, idx);
        }
        const bf16_t *xb = Xs + sl * XEL, *wb = Ws + sl * WEL;
        auto load_group = [&](int g) {
            if (VAR == 3) return;
            const int kk = g >> 1, ih = (g & 1) * 4;
            if ((g & 1) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[kk][j] = *reinterpret_cast<const bf16x8 *>(xb + j * 1024 + (kk ? xb1 : xb0));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) af[g & 1][i] = *reinterpret_cast<const bf16x8 *>(wb + (ih + i) * 1024 + (kk ? wa1 : wa0));
        };
        load_group(0);
        if (VAR != 3) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) load_group(g + 1);
            if (VAR == 5) { issue_piece(sl ^ 1, 2 * g); issue_piece(sl ^ 1, 2 * g + 1); }
            const int ih = (g & 1) * 4, kk = g >> 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][i], bfr[kk][j], acc[ih + i][j], 0, 0, 0);
            if (VAR != 3) {
                constexpr int np = VAR == 5 ? 2 : 0;
                if (g < 3) {
                    const int nrd = 4 + (((g + 1) & 1) == 0 ? 4 : 0);
#pragma unroll
                    for (int r = 0; r < nrd; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
#pragma unroll
                    for (int r = 0; r < np; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    if (nrd == 4) __builtin_amdgcn_sched_group_barrier(0x008, 12 - np, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x008, 8 - np, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < np; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 16 - np, 0);
                }
            }
        }
        sl ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += acc[i][j];
    reinterpret_cast<f32x4 *>(a.sink)[(size_t)blockIdx.x * 512 + tid] = sum;
    if (tid == 0) { a.stamps[2 * blockIdx.x] = t1 - t0; a.stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int VAR> static void run(const char *name, const Args &a, int grid, int threads)
{
    auto kern = loop_kernel<VAR>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ~2 s of back-to-back launches first: the clock the chip settles at under this load, not the boost of a cold start
    float ms = 0.f, warm = 0.f;
    while (warm < 2000.f) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 160 * 1024, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        warm += ms;
    }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 160 * 1024, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * grid);
    CK(hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc(grid), clk(grid);
    for (int i = 0; i < grid; ++i) { cyc[i] = (double)st[2 * i]; clk[i] = (double)st[2 * i] / (double)st[2 * i + 1] * 0.1; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const int waves = threads / 64;
    const double steps = (VAR == 2 ? 2.0 : 1.0) * a.nsteps;
    const double flop = (double)grid * waves * steps * 64 * 16384.0;
    printf("%-62s %7.0f cycles per step of wave 0 (%d waves per SIMD)   clock %.2f GHz   %.3f us per step   %.3f PF/s\n", name,
           cyc[grid / 2] / steps, waves / 4, clk[grid / 2], ms * 1e3 / steps, flop / (ms * 1e-3) / 1e15);
}

int main(int argc, char **argv)
{
    const int nsteps = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int grid = pr.multiProcessorCount;
    std::vector<bf16_t> h(64 * 1024);
    srand(1);
    for (auto &v : h) { const float f = (float)(rand() % 65536) / 32768.f - 1.f; uint32_t u; memcpy(&u, &f, 4); v = (bf16_t)(u >> 16); }
    Args a;
    bf16_t *src; float *sink; unsigned long long *stamps;
    CK(hipMalloc(&src, 128 * 1024)); CK(hipMalloc(&sink, (size_t)grid * 512 * 16)); CK(hipMalloc(&stamps, (size_t)grid * 16));
    CK(hipMemcpy(src, h.data(), 128 * 1024, hipMemcpyHostToDevice));
    a.src = src; a.sink = sink; a.stamps = stamps; a.nsteps = nsteps;
    printf("%d CUs, %d steps per wave; ideal 2048 cycles per step at two waves per SIMD (1024 at one)\n", grid, nsteps);
    run<0>("A  8 waves, LDS reads + MFMAs as shipped, free-running", a, grid, 512);
    run<1>("B  A + the step barrier (lgkmcnt(0); s_barrier)", a, grid, 512);
    run<2>("C  4 waves (one per SIMD), twice the steps", a, grid, 256);
    run<3>("D  8 waves, no LDS reads in the loop (MFMA issue alone)", a, grid, 512);
    run<4>("E  B + 64 KiB of LDS-DMA per step, all behind the barrier", a, grid, 512);
    run<5>("F  B + 64 KiB of LDS-DMA per step, 2 pieces per 16 MFMAs", a, grid, 512);
    return 0;
}

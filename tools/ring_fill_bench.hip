// ring_fill_bench.hip -- what one CU's LDS-DMA load path takes in, as a function of (a) how many 1-KiB DMA instructions each
// loader wave keeps in flight behind a counted `s_waitcnt vmcnt(N)` and (b) the address pattern of a 128-row x 128-byte operand
// tile: rows of a K-contiguous [N][K] matrix (row pitch 2 K bytes: the layout of the conv weights and, for 1x1 layers, of the
// NHWC activations) against one contiguous 16-KiB block per K step (a K-blocked layout).
//
// One 256-thread workgroup per CU, 4 loader waves, no consumers: the loaders never wait for a FREE word, so the figure is the
// load path's ceiling (MI355X_MICROARCH.md "ring-gemm": 68 GB/s per CU; "Indexed rows: gather into LDS": 66-73 GB/s L2-served,
// 23-24 GB/s HBM-served).  Sources: `shared` = every workgroup streams the same 2 MiB (L2-resident, the weight case),
// `private` = every workgroup streams its own 4 MiB slice of a 1 GiB buffer (HBM-served, the activation case).
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ring_fill_bench tools/ring_fill_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voffset, uint32_t soffset, void *lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voffset, soffset, 0, 0);
}

struct Args {
    const unsigned char *src;
    uint32_t bytes;        // buffer resource extent seen by a workgroup
    uint32_t wg_stride;    // byte offset between workgroups' regions (0: shared)
    uint32_t wg_mod;       // region index = blockIdx % wg_mod
    uint32_t row_stride;   // bytes between the 128 rows of a tile
    uint32_t step_stride;  // bytes between consecutive K steps' tiles
    uint32_t wrap;         // steps before the walk restarts at the region's base
    int nsteps;            // K steps (16-KiB tiles) per workgroup
    unsigned long long *cycles; // per workgroup: s_memtime ticks of the loop
};

// PIECES DMA instructions (1 KiB each) per wave and step; DEPTH = instructions left in flight behind each step's wait
template <int DEPTH, bool POLL = false>
__global__ void __launch_bounds__(256) fill_kernel(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[]; // ring: 8 slots x 16 KiB
    constexpr int PIECES = 4, NS = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane >> 3, chunk = (lane & 7) ^ lr;
    const unsigned char *base = a.src + (size_t)(blockIdx.x % a.wg_mod) * a.wg_stride;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base), 0, a.bytes, 0x00020000);
    uint32_t off[PIECES];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) off[q] = (uint32_t)(((wave + 4 * q) * 8 + lr) * a.row_stride + chunk * 16);
    const unsigned long long t0 = __builtin_readcyclecounter();
    int slot = 0;
    uint32_t so = 0, w = 0;
    for (int s = 0; s < a.nsteps; ++s) {
#pragma unroll
        for (int q = 0; q < PIECES; ++q) blds16(r, off[q], so, smem + slot * 16384 + (wave + 4 * q) * 1024);
        asm volatile("s_waitcnt vmcnt(%0)" : : "i"(DEPTH) : "memory");
        if (POLL) { // what a loader wave's flag poll does: one LDS read + `s_waitcnt lgkmcnt(0)` (does that wait drain the DMA queue?)
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(131072u + 64u) : "memory");
            if (__builtin_amdgcn_readfirstlane((int)v) == 0x12345678) so += 128; // never true; keeps the read
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        so += a.step_stride;
        if (++w == a.wrap) { w = 0; so = 0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) a.cycles[blockIdx.x] = t1 - t0;
}

template <int DEPTH, bool POLL = false> static float run(const Args &a, int grid, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fill_kernel<DEPTH, POLL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill_kernel<DEPTH, POLL>), dim3(grid), dim3(256), 160 * 1024, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((fill_kernel<DEPTH, POLL>), dim3(grid), dim3(256), 160 * 1024, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[i], e0, e1));
    }
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2];
}

int main()
{
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    const size_t big = (size_t)1 << 30;
    unsigned char *buf;
    unsigned long long *cyc;
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 1, big));
    CK(hipMalloc(&cyc, 4096 * sizeof(unsigned long long)));
    CK(hipDeviceSynchronize());
    printf("device %s, %d CUs; one 256-thread workgroup per CU, 4 loader waves, 4 DMA instructions (1 KiB) per wave and 16-KiB step\n", pr.name, ncu);
    struct Pat { const char *name; uint32_t wg_stride, wg_mod, row_stride, step_stride, wrap, bytes; };
    const Pat pats[] = {
        // shared (L2-resident): 4 n-blocks of a [512][2048] bf16 matrix = 2 MiB; every workgroup walks one n-block's 32 K steps
        {"shared  [N][K] K=2048 rows (pitch 4096 B)", 128 * 4096, 4, 4096, 128, 32, 128 * 4096},
        {"shared  [N][K] K=4608 rows (pitch 9216 B)", 128 * 9216, 4, 9216, 128, 72, 128 * 9216},
        {"shared  [N][K] pitch 4096 + 128 B (padded)", 128 * 4224, 4, 4224, 128, 32, 128 * 4224},
        {"shared  K-blocked: contiguous 16 KiB / step", 32 * 16384, 4, 128, 16384, 32, 32 * 16384},
        {"shared  [px][C] C=512 rows (pitch 1024 B)", 128 * 1024, 16, 1024, 128, 8, 128 * 1024},
        // private (HBM-served): every workgroup its own 4 MiB
        {"private [px][C] C=2048 rows (pitch 4096 B)", 4u << 20, 256, 4096, 128, 32, 4u << 20},
        {"private contiguous 16 KiB / step", 4u << 20, 256, 128, 16384, 256, 4u << 20},
    };
    const int nsteps = 512; // 8 MiB per workgroup
    for (const Pat &p : pats) {
        Args a;
        a.src = buf; a.bytes = p.bytes; a.wg_stride = p.wg_stride; a.wg_mod = p.wg_mod; a.row_stride = p.row_stride;
        a.step_stride = p.step_stride; a.wrap = p.wrap; a.nsteps = nsteps; a.cycles = cyc;
        // private + wrap: walk 4 MiB = 256 contiguous steps, or 32 steps x 8 row blocks for the pitched form
        float ms[6] = {run<0>(a, ncu, 9), run<4>(a, ncu, 9), run<8>(a, ncu, 9), run<12>(a, ncu, 9), run<16>(a, ncu, 9), run<24>(a, ncu, 9)};
        printf("%-46s", p.name);
        const int depth[6] = {0, 4, 8, 12, 16, 24};
        for (int i = 0; i < 6; ++i) {
            const double gbs = (double)nsteps * 16384 / (ms[i] * 1e-3) / 1e9; // per CU
            printf("  d%-2d %5.1f GB/s/CU", depth[i] * 4 / 4 + 0, gbs);
        }
        printf("   (KiB in flight per CU = 4 x (d + 4))\n");
    }
    // ---- few workgroups (no chip-wide HBM bound): does depth pay for HBM-served tiles, and does a flag poll's lgkmcnt(0) drain the queue? ----
    for (int grid : {8, 32}) {
        Args a;
        a.src = buf; a.bytes = 32u << 20; a.wg_stride = 32u << 20; a.wg_mod = 32; a.row_stride = 128; a.step_stride = 16384; a.wrap = 2048; a.nsteps = 2048; a.cycles = cyc;
        // each workgroup streams its own 32 MiB once, 16 KiB per step (1 GiB over 32 workgroups: nothing survives in the Infinity Cache)
        Args b = a;
        float m0 = run<0>(a, grid, 7), m8 = run<8>(a, grid, 7), m24 = run<24>(a, grid, 7), p8 = run<8, true>(a, grid, 7), p24 = run<24, true>(a, grid, 7);
        (void)b;
        auto gbs = [&](float ms) { return (double)a.nsteps * 16384 / (ms * 1e-3) / 1e9; };
        printf("%d workgroups, private contiguous 32 MiB each (HBM-served): d0 %.1f  d8 %.1f  d24 %.1f GB/s/CU;  with a flag poll (ds_read + lgkmcnt(0)) per step: d8 %.1f  d24 %.1f\n",
               grid, gbs(m0), gbs(m8), gbs(m24), gbs(p8), gbs(p24));
    }
    printf("d = DMA instructions per wave left in flight behind each step's wait; chip-wide = per-CU x %d\n", ncu);
    CK(hipFree(buf));
    CK(hipFree(cyc));
    return 0;
}

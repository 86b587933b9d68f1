// kernels_conv.hip -- the RetinaFace network's convolutions as implicit GEMM on CDNA4 matrix cores.
//
// Replaces the remote Triton forward pass of the reference (face_detection.rs:279 model_infer).
// Layout: activations NHWC bf16, weights [Cout][KH][KW][Cin] bf16, so both GEMM operands are
// K-contiguous:  D[n][m] = sum_k W[n][k] * X[m][k],  m = (b,ho,wo), k = (ky,kx,ci).
// The weight tile is the MFMA A operand and the im2col activation tile the B operand, so every lane
// ends up with 4 consecutive output CHANNELS of one pixel -> packed 8-byte NHWC stores.
//
// Per workgroup (256 threads = 4 wave64): BM x BN output tile, K step 64.  Both operand tiles go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write), double
// buffered, one barrier per K step; the LDS image is lane-linear, so the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address (and to the
// read); padding taps and rows beyond M use an out-of-range buffer offset, which reads as zeros.  v_mfma_f32_16x16x32_bf16, f32 accumulate.
// Epilogue straight from the accumulators: the weight-tile ROWS are permuted when staged so that the
// 2 x 4 accumulator registers a lane holds for an MFMA row-tile pair are 8 CONSECUTIVE output
// channels of one pixel -> one 16-byte store per lane, 64 contiguous bytes per pixel per instruction,
// no LDS round trip; the residual is prefetched before the K loop in the same layout.  Fused: +bias
// (BN folded), +residual (optionally nearest-2x upsampled), ReLU, second "BN+ReLU" output for
// pre-activation units, channel-offset stores (SSH concat), f32 + 2-class softmax for the heads.
#include "conv_device.h"

namespace rfd {


// Every wait on the vector-memory counter in this file is a FULL drain.  Rounds 1-2 shipped two counted waits (`vmcnt(XP)`: the
// youngest activation tile allowed to stay in flight behind the weight tile, in conv_igemm_kernel and conv3x3_kx_kernel); a
// build with drains everywhere measured the same end to end (7 583 / 7 615 vs 7 643 / 7 499 img/s, one box, round 3), so the
// argument about when a counted wait is sound (rule 1) no longer has to carry any kernel.
template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N == 0, "counted vmcnt waits are not used: drain");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS operand rings: NSX slots for the activation (im2col) tile, 2 for the weight tile.  With NSX = 3 the
// activation tile of step kt+2 is requested while step kt computes (HBM/L2 latency gets two steps of
// cover), the L2-hot weight tile one step ahead; the one drain per step lands both (rounds 1-2 left the youngest
// activation tile in flight behind a counted wait: no measurable difference, see wait_vmcnt).  128x128: 3*16 + 2*16 =
// 80 KiB -> two workgroups fill the CU's 160 KiB exactly.
#ifdef RFD_CLOCK_STAMPS // diagnostic build (tools/build_variant.sh clk -DRFD_CLOCK_STAMPS; tools/clock_stamps.py): the clock the chip
// holds INSIDE each kernel class while the real pass runs -- sum of s_memtime deltas (shader cycles) over sum of s_memrealtime deltas
// (100 MHz) of every workgroup, one pair of atomics per workgroup at its end (MI355X_MICROARCH.md, DVFS give-back item 6)
__device__ unsigned long long g_clock_stamps[16][2];
struct ClockStamp {
    unsigned long long t0, r0;
    int cls;
    __device__ ClockStamp(int c) : cls(c) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ ~ClockStamp()
    {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t0, dr = __builtin_amdgcn_s_memrealtime() - r0;
        if (threadIdx.x == 0) {
            atomicAdd(&g_clock_stamps[cls][0], dt);
            atomicAdd(&g_clock_stamps[cls][1], dr);
        }
    }
};
#define RFD_CLOCK(cls) ClockStamp clock_stamp__(cls)
extern "C" __attribute__((visibility("default"))) int rfd_debug_clock_stamps(unsigned long long *out, int reset)
{
    static unsigned long long host[32];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clock_stamps), sizeof host) != hipSuccess) return -1;
    for (int i = 0; i < 32; ++i) out[i] = host[i];
    if (reset) { memset(host, 0, sizeof host); if (hipMemcpyToSymbol(HIP_SYMBOL(g_clock_stamps), host, sizeof host) != hipSuccess) return -1; }
    return 0;
}
#else
#define RFD_CLOCK(cls) do { } while (0)
#endif
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSX, bool CHUNK_MAJOR = false>
__global__ void __launch_bounds__(WAVES_M *WAVES_N * 64) conv_igemm_kernel(const ConvParams p)
{
    RFD_CLOCK(0);
    constexpr int NT = WAVES_M * WAVES_N * 64, NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    // staging: one DMA instruction = 8 rows x 128 B; wave w issues pieces w, w+NW, ...
    constexpr int XP = BM / 8 / NW, WP = (BN / 8 + NW - 1) / NW;
    static_assert((BM / 8) % NW == 0, "X tile pieces must divide over the waves");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem);             // [slots][BM*64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int HoWo = p.Ho * p.Wo;
    const int M = p.B * HoWo;
    const int K1 = p.KH * p.KW * p.Cin;
    const int K = K1 + p.Cin2; // second K segment: the fused 1x1 shortcut conv
    const int nk1 = K1 >> 6, nk = K >> 6;
    bf16_t *Ws = Xs + (nk > 1 ? NSX : 1) * BM * 64;            // [2][BN*64]; one slot each for a single K step
    const int tiles_n = p.Cout / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

    // ---- per-lane im2col bookkeeping: lane (r = lane/8, slot = lane%8) of piece q stages LDS row
    //      R = (wave + NW*q)*8 + r, slot `slot`, which holds global chunk slot ^ r of that row.
    //      Offsets are BYTES relative to the tensor base (32-bit: tensors are < 4 GiB). ----
    const int lr = lane >> 3, chunk = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)p.B * p.H * p.W * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.w), 0, (uint32_t)((size_t)p.Cout * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.Cin2 ? p.x2 : p.x), 0,
        (uint32_t)(p.Cin2 ? (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2 : 0), 0x00020000);
    uint32_t xoff[XP], xoff2[XP];
    int hi0[XP], wi0[XP];
#pragma unroll
    for (int q = 0; q < XP; ++q) {
        const int m = m0 + (wave + NW * q) * 8 + lr;
        xoff2[q] = kOob;
        if (m < M) {
            const int b = m / HoWo, rem = m - b * HoWo;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0[q] = ho * p.stride - p.pad;
            wi0[q] = wo * p.stride - p.pad;
            // may be "negative" (wraps) for padded rows/cols; adding a valid tap brings it back in range
            xoff[q] = (uint32_t)(((((long long)b * p.H + hi0[q]) * p.W + wi0[q]) * p.ldx + p.x_coff + chunk * 8) * 2);
            if (p.Cin2)
                xoff2[q] = (uint32_t)(((((long long)b * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.Cin2 + chunk * 8) * 2);
        } else {
            hi0[q] = -(1 << 28); // fails every bounds check -> zero rows
            wi0[q] = 0;
            xoff[q] = 0;
        }
    }
    uint32_t woff[WP];
#pragma unroll
    for (int q = 0; q < WP; ++q) {
        const int piece = wave + NW * q;
        // LDS row rho = i*16 + fq*4 + r (the MFMA A-operand row) holds output channel
        // (i>>1)*32 + fq*8 + (i&1)*4 + r of the wave's WN-wide slice
        const int rho = (piece < BN / 8 ? piece * 8 + lr : 0);
        const int rw_ = rho % WN, i_ = rw_ >> 4, fq_ = (rw_ >> 2) & 3, r_ = rw_ & 3;
        const int chn = (rho - rw_) + (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_;
        woff[q] = (uint32_t)(((size_t)(n0 + chn) * K + chunk * 8) * 2);
    }

    int ky = 0, kx = 0, kc = 0, kt_x = 0, kt_w = 0; // positions of the NEXT tiles to stage
    const int kc_n = p.Cin >> 6;
    auto stage_x = [&](int slot) {
        if (kt_x < nk1) {
            const uint32_t tap = (uint32_t)((ky * p.W + kx) * p.ldx * 2); // scalar
#pragma unroll
            for (int q = 0; q < XP; ++q) {
                const bool ok = (unsigned)(hi0[q] + ky) < (unsigned)p.H && (unsigned)(wi0[q] + kx) < (unsigned)p.W;
                blds16(rx, ok ? xoff[q] + tap : kOob, CHUNK_MAJOR ? (uint32_t)__builtin_amdgcn_readfirstlane(kc << 7) : (uint32_t)(kc << 7),
                       Xs + slot * BM * 64 + (wave + NW * q) * 512);
            }
        } else {
#pragma unroll
            for (int q = 0; q < XP; ++q)
                blds16(rx2, xoff2[q], (uint32_t)((kt_x - nk1) << 7), Xs + slot * BM * 64 + (wave + NW * q) * 512);
        }
        ++kt_x;
        // K order: (ky, kx, chunk) as the weight rows are laid out -- or, for the layer shapes the halo-tile 3x3 kernel accepts
        // (CHUNK_MAJOR instantiation, picked by launch_conv), chunk-major (chunk, ky, kx), the order that kernel and the merged-kx
        // kernel are bound to: every kernel that can run such a layer gives bit-identical results, whatever batch size picked
        // it.  A template parameter, not a runtime flag: the extra scalar state of the second order cost this kernel its
        // register budget (scratch spills, whose reloads drain the DMA counter: 2x slower on every layer).
        if (CHUNK_MAJOR) {
            if (++kx == p.KW) { kx = 0; if (++ky == p.KH) { ky = 0; ++kc; } }
        } else if (++kc == kc_n) {
            kc = 0;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    };
    int wtap = 0, wkc = 0; // CHUNK_MAJOR: tap / chunk of the next weight tile
    auto stage_w = [&](int slot) {
        // (readfirstlane: the value is wave-uniform, but left in a VGPR the compiler wraps every DMA in a waterfall loop)
        const uint32_t col = CHUNK_MAJOR ? (uint32_t)__builtin_amdgcn_readfirstlane(kt_w < nk1 ? (wtap * p.Cin + (wkc << 6)) * 2 : kt_w << 7)
                                         : (uint32_t)(kt_w << 7);
#pragma unroll
        for (int q = 0; q < WP; ++q) {
            const int piece = wave + NW * q;
            if (piece < BN / 8) blds16(rw, woff[q], col, Ws + slot * BN * 64 + piece * 512);
        }
        ++kt_w;
        if (CHUNK_MAJOR && ++wtap == p.KH * p.KW) { wtap = 0; ++wkc; }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    constexpr int TH = TN / 2; // 8-channel groups per lane and pixel
    uint4 resv[TM][TH];
    conv_prefetch_residual<TM, TH, WM, WN>(p, resv, m0, n0, wm, wn, frow, fq, M, HoWo);

    // optional input affine (+ReLU): per-channel scale/shift staged once in LDS behind the operand slots
    float *Sc = reinterpret_cast<float *>(Ws + (nk > 1 ? 2 : 1) * BN * 64);
    if (p.in_scale) {
        for (int c = tid; c < K1; c += NT) {
            Sc[c] = p.in_scale[c];
            Sc[K1 + c] = p.in_shift[c];
        }
        __syncthreads();
    }
    // prologue, in queue order: X0, W0, (X1)
    stage_x(0);
    stage_w(0);
    if (NSX > 2 && nk > 1) stage_x(1);

    int xslot = 0, wslot = 0, xstage = NSX - 1, wstage = 1;
    for (int kt = 0; kt < nk; ++kt) {
        // step kt needs X(kt), W(kt); the drain also lands X(kt+1) (3-slot ring), requested a whole step ago
        wait_vmcnt<0>();
        // ... in every wave; the same barrier frees the slots consumed in step kt-1 for restaging
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* no LDS read in flight at a barrier that frees a ring slot for DMA (tools/isa_check.py) */
        if (kt + 1 < nk) {
            stage_w(wstage);
            wstage ^= 1;
        }
        if (kt + NSX - 1 < nk) {
            stage_x(xstage);
            xstage = xstage + 1 == NSX ? 0 : xstage + 1;
        }
        const int slot = xslot;
        xslot = xslot + 1 == NSX ? 0 : xslot + 1;
        const bf16_t *xs = Xs + slot * BM * 64 + (wm * WM) * 64;
        const bf16_t *ws = Ws + wslot * BN * 64 + (wn * WN) * 64;
        wslot ^= 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[TN], bfr[TM];
            const int ch = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int r = i * 16 + frow;
                af[i] = *reinterpret_cast<const bf16x8 *>(ws + r * 64 + ((ch ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int r = j * 16 + frow;
                bfr[j] = *reinterpret_cast<const bf16x8 *>(xs + r * 64 + ((ch ^ (r & 7)) << 3));
            }
            if (p.in_scale) {
                // this lane's 8 operand elements are input channels kt*64 + kk*32 + fq*8 .. +7 of one pixel
                const float *sc = Sc + kt * 64 + kk * 32 + fq * 8;
                const float4 s0 = *reinterpret_cast<const float4 *>(sc), s1 = *reinterpret_cast<const float4 *>(sc + 4);
                const float4 t0 = *reinterpret_cast<const float4 *>(sc + K1), t1 = *reinterpret_cast<const float4 *>(sc + K1 + 4);
                const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float tt[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    bf16x8 v = bfr[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)fmaxf(__builtin_fmaf((float)v[e], ss[e], tt[e]), 0.f); // one fused multiply-add: the same in every kernel that applies the input affine
                    bfr[j] = v;
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    conv_epilogue<TM, TN, WM, WN>(p, acc, resv, m0, n0, wm, wn, frow, fq, M);
}

// ------------------------------------------------------------------------------------------------
// Pointwise streaming kernel for the SHORT-K, WIDE-N 1x1 layers (conv3 of the bottleneck units of stages 2-3:
// K = 128 / 256 -> N = 512 / 1024, + residual, one or two outputs).  These layers are memory-bound (1.7 KiB of residual +
// outputs per pixel against 2-8 K-steps of MFMA work), and with one short-lived workgroup per 128 x 128 tile the CU's
// memory pipeline idles through every tile's prologue (address set-up, first-operand latency), epilogue (bias fetch, store
// drain) and the workgroup relaunch: 3.1 TB/s of HBM-side traffic where the tensors need 5+.
// Here a workgroup is PERSISTENT and X-STATIONARY: it owns a stripe of 128-pixel tiles; per tile the whole activation tile
// [128][K] is staged once (instead of once per N tile: -25 % bytes through the CU's load path), then the chunks of 128 output
// channels follow one another without a break.  Software pipeline, per chunk c:
//     K-steps of c (MFMA; the weight K-steps of c+1 are issued meanwhile, one per step, a whole chunk ahead)
//     s_waitcnt vmcnt(0)        <- the ONLY wait: stores of c-1, residual of c, weights of c+1 -- all issued >= 1 step ago
//     epilogue of c: request the residual of c+1 (registers, two alternating sets), + bias (LDS table), store c
// so the stores of chunk c and the residual of c+1 travel while the MFMAs of c+1 run, and what the wait really waits for
// is the HBM pipe itself.  Every wait is a full drain on purpose: a first version with COUNTED waits returned stale
// activation tiles in the first chunk of ~3 % of the tiles (tests/test_persistent_gpu.py).  It had the previous chunk's output
// STORES in flight under its counted waits, and on this ISA loads and stores share vmcnt but retire out of order with respect
// to each other -- the documented case in which only vmcnt(0) is sound (rule 1 as reworded in round 3;
// round 2 had blamed L2-served DMA overtaking HBM-served DMA).  The residual loads are inline asm (hipcc would otherwise
// place its own wait at an unknown point).  8 waves: 4 (pixels) x 2 (channels), 32 x 64 outputs each.
// LDS: activation tile NK x 16 KiB + weight ring (NK + 1) x 16 KiB + 12 B per output channel: 156 KiB at K = 256.
// ------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 make_srd(const void *base, uint32_t bytes)
{
    const uint64_t a = (uint64_t)base;
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
// one 16-byte buffer load the compiler does not count (the destination is valid only after the caller's own wait)
__device__ __forceinline__ void asm_buffer_load_b128(u32x4 &dst, uint32_t voff, const u32x4 &srd)
{
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(srd) : "memory");
}

// 8 consecutive floats of an LDS table as four 64-bit reads.  NOT ds_read_b128: lanes that share an address (here the
// 16 lanes of a quarter wave) reading 128 bits is the access shape that returned wrong data in lanes 48-63 whenever MFMA
// waves of ANOTHER kernel shared the CU -- and this kernel does co-reside with the other chain's
// convolution workgroups in the split mode.
__device__ __forceinline__ void lds_table_read8(const float *tab, float (&v)[8])
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a, b, c, d;
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
                 : "v"((uint32_t)(uintptr_t)tab)
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = d.x; v[7] = d.y;
}

// Two / three tables at once behind ONE wait (round 4): the epilogues of the pointwise kernels read bias, scale and shift for the
// same eight channels; as three lds_table_read8 calls that was three serialised LDS round trips per 8-channel group with every
// wave of the workgroup in its epilogue at the same time (nothing else running on the CU): one round trip now.
__device__ __forceinline__ void lds_table_read8x2(const float *t0, const float *t1, float (&a)[8], float (&b)[8])
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 r[8];
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                 "ds_read_b64 %4, %9\n\tds_read_b64 %5, %9 offset:8\n\tds_read_b64 %6, %9 offset:16\n\tds_read_b64 %7, %9 offset:24\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"((uint32_t)(uintptr_t)t0), "v"((uint32_t)(uintptr_t)t1)
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[2 * k] = r[k].x; a[2 * k + 1] = r[k].y; b[2 * k] = r[4 + k].x; b[2 * k + 1] = r[4 + k].y; }
}
__device__ __forceinline__ void lds_table_read8x3(const float *t0, const float *t1, const float *t2, float (&a)[8], float (&b)[8], float (&c)[8])
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 r[12];
    asm volatile("ds_read_b64 %0, %12\n\tds_read_b64 %1, %12 offset:8\n\tds_read_b64 %2, %12 offset:16\n\tds_read_b64 %3, %12 offset:24\n\t"
                 "ds_read_b64 %4, %13\n\tds_read_b64 %5, %13 offset:8\n\tds_read_b64 %6, %13 offset:16\n\tds_read_b64 %7, %13 offset:24\n\t"
                 "ds_read_b64 %8, %14\n\tds_read_b64 %9, %14 offset:8\n\tds_read_b64 %10, %14 offset:16\n\tds_read_b64 %11, %14 offset:24\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
                   "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11])
                 : "v"((uint32_t)(uintptr_t)t0), "v"((uint32_t)(uintptr_t)t1), "v"((uint32_t)(uintptr_t)t2)
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[2 * k] = r[k].x; a[2 * k + 1] = r[k].y;
        b[2 * k] = r[4 + k].x; b[2 * k + 1] = r[4 + k].y;
        c[2 * k] = r[8 + k].x; c[2 * k + 1] = r[8 + k].y;
    }
}

template <int NK, bool HAS_Y, bool HAS_Y2>
__global__ void __launch_bounds__(512) pw_stream_kernel(const ConvParams p)
{
    RFD_CLOCK(1);
    constexpr int BM = 128, WSLOTS = NK + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem);   // [NK][BM*64]
    bf16_t *Ws = Xs + NK * BM * 64;                  // [WSLOTS][128*64]
    float *Tab = reinterpret_cast<float *>(Ws + WSLOTS * 128 * 64); // bias [N] | scale2 [N] | shift2 [N]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int M = p.B * p.Ho * p.Wo, K = p.Cin, N = p.Cout;
    const int NC = N >> 7;                           // output chunks of 128 channels (even)
    const int tiles_m = (M + BM - 1) / BM;
    const int lr = lane >> 3, chunk = (lane & 7) ^ lr, frow = lane & 15, fq = lane >> 4;
    if ((int)blockIdx.x >= tiles_m) return;

    for (int c = tid; c < N; c += 512) {
        Tab[c] = p.bias[c];
        if (HAS_Y2) { Tab[N + c] = p.scale2[c]; Tab[2 * N + c] = p.shift2[c]; }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w), 0, (uint32_t)((size_t)N * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(HAS_Y ? p.y : p.y2, 0, (uint32_t)((size_t)M * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry2 = __builtin_amdgcn_make_buffer_rsrc(p.y2 ? p.y2 : p.y, 0, (uint32_t)((size_t)M * N * 2), 0x00020000);
    const u32x4 rres = make_srd(p.res, (uint32_t)((size_t)M * N * 2));
    // weight pieces of a K-step (16 x [8 rows x 128 B]): wave w stages pieces w and w + 8; LDS row rho holds output channel
    // perm(rho) of the chunk, so that a lane's accumulators are 8 consecutive channels (see conv_igemm_kernel)
    uint32_t woff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rho = (wave + 8 * q) * 8 + lr;
        const int rw_ = rho & 63, i_ = rw_ >> 4, fq_ = (rw_ >> 2) & 3, r_ = rw_ & 3;
        const int chn = (rho - rw_) + (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_;
        woff[q] = (uint32_t)(((size_t)chn * K + chunk * 8) * 2);
    }
    // the weight stream: position of the NEXT step to issue (cycles through all chunks, tile after tile), one chunk ahead
    int wi_nc = 0, wi_kt = 0, wi_slot = 0;
    auto issue_w = [&]() {
        const uint32_t so = (uint32_t)((((size_t)wi_nc * 128) * K + (wi_kt << 6)) * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) blds16(rw, woff[q], so, Ws + wi_slot * 128 * 64 + (wave + 8 * q) * 512);
        if (++wi_kt == NK) { wi_kt = 0; if (++wi_nc == NC) wi_nc = 0; }
        wi_slot = wi_slot + 1 == WSLOTS ? 0 : wi_slot + 1;
    };
    // this lane's output position inside a (tile, chunk): pixel rows wm*32 + j*16 + frow, channels wn*64 + h*32 + fq*8 .. +7
    const uint32_t lane_off = (uint32_t)((((size_t)(wm * 32 + frow)) * N + wn * 64 + fq * 8) * 2);
    auto issue_res = [&](u32x4 (&r)[4], int mt, int nc) { // 4 loads, always issued (out of range -> zeros)
        const bool tile_ok = mt < tiles_m;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = tile_ok && mt * BM + wm * 32 + j * 16 + frow < M;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                asm_buffer_load_b128(r[j * 2 + h], ok ? lane_off + (uint32_t)((((size_t)mt * BM + j * 16) * N + (nc << 7) + h * 32) * 2) : kOob, rres);
        }
    };

    u32x4 resA[4], resB[4];
    issue_res(resA, blockIdx.x, 0);
#pragma unroll
    for (int i = 0; i < NK; ++i) issue_w(); // the first chunk's weights
    int cslot = 0; // ring slot of the weight step consumed next

    for (int mt = blockIdx.x; mt < tiles_m; mt += gridDim.x) {
        const int m0 = mt * BM;
        // every wave is done with the previous tile's activation tile -- including fragment reads the compiler sank below their
        // MFMAs: an LDS read still in flight at the barrier would race with the DMA that refills its slot
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // activation tile: NK K-steps x 16 pieces; wave w stages pieces w and w + 8 of every step
#pragma unroll
        for (int kt = 0; kt < NK; ++kt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + (wave + 8 * q) * 8 + lr;
                blds16(rx, m < M ? (uint32_t)(((size_t)m * K + chunk * 8) * 2) : kOob, (uint32_t)(kt << 7),
                       Xs + kt * BM * 64 + (wave + 8 * q) * 512);
            }
        wait_vmcnt<0>(); // the activation tile (this chunk's weights landed at the previous drain, or are older)
        auto do_chunk = [&](int nc, u32x4 (&cur)[4], u32x4 (&nxt)[4]) __attribute__((always_inline)) {
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                // everything this step reads was drained by every wave before it got here; the barrier publishes it and
                // frees the ring slot read one step ago for the matching K-step of the NEXT chunk (lgkmcnt(0): hipcc sinks the
                // last ds_read pair of step kt-1 below that step's MFMAs, and a bare barrier would let issue_w() overwrite the
                // slot while they are still in flight -- a write-after-read race found in the round-2 ISA)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue_w();
                const bf16_t *xs = Xs + kt * BM * 64 + (wm * 32) * 64;
                const bf16_t *ws = Ws + cslot * 128 * 64 + (wn * 64) * 64;
                cslot = cslot + 1 == WSLOTS ? 0 : cslot + 1;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 af[4], bfr[2];
                    const int ch = kk * 4 + fq;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = i * 16 + frow;
                        af[i] = *reinterpret_cast<const bf16x8 *>(ws + r * 64 + ((ch ^ (r & 7)) << 3));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int r = j * 16 + frow;
                        bfr[j] = *reinterpret_cast<const bf16x8 *>(xs + r * 64 + ((ch ^ (r & 7)) << 3));
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            }
            // ---- the one wait of the chunk: residual of this chunk, weights of the next, stores of the previous ----
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- epilogue: request the NEXT chunk's residual (next tile's chunk 0 after the last chunk), then this one's ----
            if (nc + 1 < NC) issue_res(nxt, mt, nc + 1);
            else issue_res(nxt, mt + (int)gridDim.x, 0);
            const int n0 = nc << 7;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n0 + wn * 64 + h * 32 + fq * 8;
                float bias[8], s2[8], t2[8];
                if (HAS_Y2) lds_table_read8x3(Tab + n, Tab + N + n, Tab + 2 * N + n, bias, s2, t2);
                else lds_table_read8(Tab + n, bias);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int m = m0 + wm * 32 + j * 16 + frow;
                    const uint32_t off = m < M ? (uint32_t)(((size_t)m * N + n) * 2) : kOob;
                    const u32x4 rv = cur[j * 2 + h];
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = acc[2 * h][j][k] + bias[k];
                        v[4 + k] = acc[2 * h + 1][j][k] + bias[4 + k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] += bf16_bits_to_f32(rv[k] & 0xffffu);
                        v[2 * k + 1] += bf16_bits_to_f32(rv[k] >> 16);
                    }
                    if (HAS_Y) {
                        float o[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = p.relu ? fmaxf(v[k], 0.f) : v[k];
                        const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry, off, 0, 0);
                    }
                    if (HAS_Y2) {
                        float o[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = fmaxf(v[k] * s2[k] + t2[k], 0.f);
                        const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry2, off, 0, 0);
                    }
                }
            }
        };
        for (int nc = 0; nc < NC; nc += 2) { // two chunks per turn: the residual register sets alternate without copies
            do_chunk(nc, resA, resB);
            do_chunk(nc + 1, resB, resA);
        }
    }
    wait_vmcnt<0>(); // the weight steps and residual loads issued beyond the end
}

// ------------------------------------------------------------------------------------------------
// pw_stream with the NEXT unit's conv1 run back to back on the tile it has just produced (round 3; stage 2's dim-match units:
// conv3 128 -> 512 + residual, then conv1 512 -> 128 of the following unit).  Per 128-pixel tile and 128-channel chunk c of the
// raw sum the epilogue, besides storing the chunk, writes relu(raw * scale + shift) -- the producer unit's BN + ReLU, on the
// bf16-rounded raw value exactly as pw_gemm applies it to its landed tile -- as a bf16 MFMA operand tile [128 px][128 ch] into
// LDS; two more K steps then accumulate W1[:, chunk c] x that tile into the conv1 accumulators, which live across the chunks.
// The raw tensor is therefore written once and read once (as the next residual), not twice, and one launch disappears.
// Same K order and MFMA sequence per output as pw_stream + pw_gemm: bit-identical results.
// One weight stream carries both layers' 16-KiB steps, per chunk W3(c, 0..NK-1), W1(c, 0), W1(c, 1), a whole chunk ahead in an
// (NK + 3)-slot ring; the one drain per chunk sits where pw_stream has it (between the conv3 steps and the epilogue).
// LDS at K3 = 128: X 32 + operand tile 32 + ring 80 KiB + tables 6.5 KiB.
// ------------------------------------------------------------------------------------------------
// ACT_OUT (the LAST unit of a stage: stage 1 -> stage 2's first conv1, 256 -> 128 at 160 x 160): the unit stores its BN + ReLU
// output (y2, pw_stream's arithmetic: v * scale + shift, ReLU) instead of the raw sum, and that stored value IS conv1's operand.
template <int NK, bool ACT_OUT>
__global__ void __launch_bounds__(512) pw_b2b_kernel(const ConvParams p)
{
    RFD_CLOCK(2);
    constexpr int BM = 128, S = NK + 2, WSLOTS = S + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem);   // [NK][BM*64]
    bf16_t *As = Xs + NK * BM * 64;                  // [2][BM*64]: the activated chunk as conv1's operand (two K steps)
    bf16_t *Ws = As + 2 * BM * 64;                   // [WSLOTS][128*64]
    float *Tab = reinterpret_cast<float *>(Ws + WSLOTS * 128 * 64); // bias3 [N] | scale [N] | shift [N] | bias1 [128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int M = p.B * p.Ho * p.Wo, K = p.Cin, N = p.Cout, N1 = 128;
    const int NC = N >> 7;
    const int tiles_m = (M + BM - 1) / BM;
    const int lr = lane >> 3, chunk = (lane & 7) ^ lr, frow = lane & 15, fq = lane >> 4;
    if ((int)blockIdx.x >= tiles_m) return;

    for (int round = 0; round < (N + 511) / 512; ++round) { // scalar trip count, predicated body (tools/isa_check.py can follow EXEC)
        const int c = tid + round * 512;
        if (c < N) {
            Tab[c] = p.bias[c];
            Tab[N + c] = p.scale2[c];
            Tab[2 * N + c] = p.shift2[c];
        }
        if (c < N1) Tab[3 * N + c] = p.bias1[c];
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)M * K * 2), 0x00020000);
    // ONE resource for both filter banks (both live in the network's weight buffer, W1 behind W3: the launcher checks): a
    // select between two descriptors made hipcc park them in scratch and wrap every DMA in a waterfall loop
    const uint32_t w1_delta = (uint32_t)((const char *)p.w1 - (const char *)p.w);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w), 0, w1_delta + (uint32_t)((size_t)N1 * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(ACT_OUT ? p.y2 : p.y, 0, (uint32_t)((size_t)M * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt1 = __builtin_amdgcn_make_buffer_rsrc(p.t1, 0, (uint32_t)((size_t)M * N1 * 2), 0x00020000);
    const u32x4 rres = make_srd(p.res, (uint32_t)((size_t)M * N * 2));
    // weight pieces of a step (16 x [8 rows x 128 B]): wave w stages pieces w and w + 8; LDS row rho holds output channel perm(rho)
    // of the 128-row block (8 consecutive channels per lane in the epilogues); row pitch K for W3, N for W1
    uint32_t woff[2], woff1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rho = (wave + 8 * q) * 8 + lr;
        const int rw_ = rho & 63, i_ = rw_ >> 4, fq_ = (rw_ >> 2) & 3, r_ = rw_ & 3;
        const int chn = (rho - rw_) + (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_;
        woff[q] = (uint32_t)(((size_t)chn * K + chunk * 8) * 2);
        woff1[q] = (uint32_t)(((size_t)chn * N + chunk * 8) * 2);
    }
    // the unified weight stream: position of the NEXT step to issue; per chunk W3(c, 0..NK-1), W1(c, 0), W1(c, 1); one chunk ahead
    int wi_nc = 0, wi_s = 0, wi_slot = 0;
    auto issue_w = [&]() {
        bf16_t *dst = Ws + wi_slot * 128 * 64;
        const bool w3 = wi_s < NK; // wave-uniform
        // W3: rows of chunk wi_nc, columns of K step wi_s; W1: all 128 rows, columns = the channels of chunk wi_nc, half wi_s - NK
        const uint32_t so = w3 ? (uint32_t)((((size_t)wi_nc * 128) * K + (wi_s << 6)) * 2) : w1_delta + (uint32_t)(((wi_nc << 7) + ((wi_s - NK) << 6)) * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) blds16(rw, w3 ? woff[q] : woff1[q], (uint32_t)__builtin_amdgcn_readfirstlane(so), dst + (wave + 8 * q) * 512);
        if (++wi_s == S) { wi_s = 0; if (++wi_nc == NC) wi_nc = 0; }
        wi_slot = wi_slot + 1 == WSLOTS ? 0 : wi_slot + 1;
    };
    const uint32_t lane_off = (uint32_t)((((size_t)(wm * 32 + frow)) * N + wn * 64 + fq * 8) * 2);
    auto issue_res = [&](u32x4 (&r)[4], int mt, int nc) {
        const bool tile_ok = mt < tiles_m;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = tile_ok && mt * BM + wm * 32 + j * 16 + frow < M;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                asm_buffer_load_b128(r[j * 2 + h], ok ? lane_off + (uint32_t)((((size_t)mt * BM + j * 16) * N + (nc << 7) + h * 32) * 2) : kOob, rres);
        }
    };

    u32x4 resA[4], resB[4];
    issue_res(resA, blockIdx.x, 0);
#pragma unroll
    for (int i = 0; i < S; ++i) issue_w(); // the first chunk's steps
    int cslot = 0; // ring slot of the step consumed next

    // one K step of either GEMM on this wave's 32 x 64 tile: A = ring slot rows wn*64.., B = 128-row operand tile rows wm*32..
    auto mma_step = [&](f32x4 (&acc)[4][2], const bf16_t *btile) __attribute__((always_inline)) {
        const bf16_t *xs = btile + (wm * 32) * 64;
        const bf16_t *ws = Ws + cslot * 128 * 64 + (wn * 64) * 64;
        cslot = cslot + 1 == WSLOTS ? 0 : cslot + 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[2];
            const int ch = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 16 + frow;
                af[i] = *reinterpret_cast<const bf16x8 *>(ws + r * 64 + ((ch ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = j * 16 + frow;
                bfr[j] = *reinterpret_cast<const bf16x8 *>(xs + r * 64 + ((ch ^ (r & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    for (int mt = blockIdx.x; mt < tiles_m; mt += gridDim.x) {
        const int m0 = mt * BM;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // every wave is done with the previous tile's tiles
#pragma unroll
        for (int kt = 0; kt < NK; ++kt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + (wave + 8 * q) * 8 + lr;
                blds16(rx, m < M ? (uint32_t)(((size_t)m * K + chunk * 8) * 2) : kOob, (uint32_t)(kt << 7), Xs + kt * BM * 64 + (wave + 8 * q) * 512);
            }
        wait_vmcnt<0>(); // the activation tile (+ the previous tile's t1 stores)
        f32x4 acc1[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto do_chunk = [&](int nc, u32x4 (&cur)[4], u32x4 (&nxt)[4]) __attribute__((always_inline)) {
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) { // conv3 steps of this chunk
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue_w();
                mma_step(acc, Xs + kt * BM * 64);
            }
            // ---- the one drain of the chunk: residual of this chunk, W1 steps of this chunk, stores of the previous one ----
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (nc + 1 < NC) issue_res(nxt, mt, nc + 1);
            else issue_res(nxt, mt + (int)gridDim.x, 0);
            const int n0 = nc << 7;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n0 + wn * 64 + h * 32 + fq * 8;
                float bias[8], sc[8], sh[8];
                lds_table_read8x3(Tab + n, Tab + N + n, Tab + 2 * N + n, bias, sc, sh);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wm * 32 + j * 16 + frow, m = m0 + row;
                    const uint32_t off = m < M ? (uint32_t)(((size_t)m * N + n) * 2) : kOob;
                    const u32x4 rv = cur[j * 2 + h];
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = acc[2 * h][j][k] + bias[k];
                        v[4 + k] = acc[2 * h + 1][j][k] + bias[4 + k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] += bf16_bits_to_f32(rv[k] & 0xffffu);
                        v[2 * k + 1] += bf16_bits_to_f32(rv[k] >> 16);
                    }
                    uint2 alo, ahi;
                    if (ACT_OUT) { // the stage output relu(v * scale + shift) (pw_stream's y2), stored and used as the operand
                        float a[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) a[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
                        alo = pack_bf16x4(a[0], a[1], a[2], a[3]); ahi = pack_bf16x4(a[4], a[5], a[6], a[7]);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{alo.x, alo.y, ahi.x, ahi.y}, ry, off, 0, 0);
                    } else {
                        const uint2 lo = pack_bf16x4(v[0], v[1], v[2], v[3]), hi = pack_bf16x4(v[4], v[5], v[6], v[7]);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry, off, 0, 0);
                        // conv1's operand: relu(fma(raw as stored, scale, shift)) -- pw_gemm's input affine, on the same bf16 values
                        const uint32_t rb[4] = {lo.x, lo.y, hi.x, hi.y};
                        float a[8];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            a[2 * k] = fmaxf(__builtin_fmaf(bf16_bits_to_f32(rb[k] & 0xffffu), sc[2 * k], sh[2 * k]), 0.f);
                            a[2 * k + 1] = fmaxf(__builtin_fmaf(bf16_bits_to_f32(rb[k] >> 16), sc[2 * k + 1], sh[2 * k + 1]), 0.f);
                        }
                        alo = pack_bf16x4(a[0], a[1], a[2], a[3]); ahi = pack_bf16x4(a[4], a[5], a[6], a[7]);
                    }
                    // K step wn of the operand tile (channels wn*64 .. +63 of the chunk), 16-byte slot h*4 + fq of pixel row `row`
                    *reinterpret_cast<uint4 *>(As + wn * BM * 64 + row * 64 + (((h * 4 + fq) ^ (row & 7)) << 3)) = make_uint4(alo.x, alo.y, ahi.x, ahi.y);
                }
            }
#pragma unroll
            for (int k1 = 0; k1 < 2; ++k1) { // conv1 steps on the chunk just activated (the barrier publishes the operand tile)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue_w();
                mma_step(acc1, As + k1 * BM * 64);
            }
        };
        for (int nc = 0; nc < NC; nc += 2) {
            do_chunk(nc, resA, resB);
            do_chunk(nc + 1, resB, resA);
        }
        // ---- conv1 epilogue: t1 = relu(acc1 + bias1), 16-byte stores (8 consecutive channels of one pixel per lane) ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = wn * 64 + h * 32 + fq * 8;
            float b1[8];
            lds_table_read8(Tab + 3 * N + n, b1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m0 + wm * 32 + j * 16 + frow;
                float o[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = fmaxf(acc1[2 * h][j][k] + b1[k], 0.f);
                    o[4 + k] = fmaxf(acc1[2 * h + 1][j][k] + b1[4 + k], 0.f);
                }
                const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, rt1, m < M ? (uint32_t)(((size_t)m * N1 + n) * 2) : kOob, 0, 0);
            }
        }
    }
    wait_vmcnt<0>(); // the weight steps and residual loads issued beyond the end
}

// ------------------------------------------------------------------------------------------------
// The same pair (a unit's conv3 + residual, then the next unit's conv1) with the tile cut the other way: every wave owns 16 PIXELS
// and all channels.  Its activation tile [16 px][K3] lives in registers as MFMA B fragments (loaded once per tile straight from
// global memory), and the epilogue's lane layout -- pixel = lane % 16, eight consecutive channels per lane and 32-channel group
// -- IS the B-fragment layout of conv1's K slices, so the activated chunk feeds the second GEMM from registers: no activation
// tile and no operand tile in LDS at all.  LDS holds only the weight ring, which therefore has room for a whole chunk of BOTH
// layers even at K3 = 256: 8 slot-steps of 16 KiB + 1 = 144 KiB (pw_b2b_kernel's layout needs 64 + 32 KiB of tiles there and
// cannot).  Price: every wave reads every weight row from LDS (one fragment read per MFMA instead of one per two): the kernel is
// LDS-read bound at about the time its HBM bytes take, which is what these layers are bound by anyway.
// Per slot-step [128 rows][64 k]: 16 fragment reads, 16 MFMAs per wave; one barrier; the stream a whole chunk ahead; one drain
// per chunk.  Same K order per output as pw_stream + pw_gemm: bit-identical.  NK = K3 / 64, N1B = N1 / 128.
// ------------------------------------------------------------------------------------------------
// NK2 > 0: the first unit of a stage -- the 1x1 stride-2 shortcut conv rides in the same GEMM as a second K segment (NK2 more K
// steps whose activation fragments are gathered from x2 at (2 ho, 2 wo)), its bias is added to conv3's, and there is no residual.
// RESIDENT (NCR = number of chunks, > 0): both filter banks fit LDS (stage 1 -> 2 boundary: 32 + 64 KiB) and are staged ONCE, in
// stream order; after that there is no weight traffic, no ring and therefore no barrier: every wave is an independent pipeline over
// its 16 pixels.
// What bounds the pair kernels (round 4, corrected): NOT the CU's load path -- tools/ring_fill_bench.hip measures 95-125 GB/s per CU
// for L2-served tiles even with a single 16-KiB step in flight, where round 3 had read the kernels' 25 GB/s per CU as a ceiling.
// Phase stamps (-DRFD_PAIR_STAMPS, tools/pair_stamps.py) on stage 3's <4, 2>: fragment reads + MFMAs 46 % of a wave's lifetime
// (600-1000 cycles per 16-MFMA step: two waves share a SIMD's matrix pipe), the chunk epilogues 27 % (VALU: ~9 instructions per
// output element, every wave in its epilogue at the same time, matrix pipe idle), barriers 16 %, weight-DMA issue 9 %, drains 2 %.
// PMC: 45 % of wave cycles waiting, MFMA busy 18 %, LDS 22 %, no bank conflicts (profiles/r04_pmc_sq_counters_per_kernel.txt).
#ifdef RFD_PAIR_STAMPS // diagnostic build (tools/build_variant.sh ... -DRFD_PAIR_STAMPS): where a wave's cycles go, summed over the grid
__device__ unsigned long long g_pair_prof[4096 * 10]; // [workgroup * 8 + wave][phase]: plain stores, no atomics
#define RFD_STAMP(i) do { const unsigned long long t__ = __builtin_readcyclecounter(); prof[i] += t__ - tlast; tlast = t__; } while (0)
extern "C" __attribute__((visibility("default"))) int rfd_debug_pair_prof(unsigned long long *out, int reset)
{
    static unsigned long long host[4096 * 10];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pair_prof), sizeof host) != hipSuccess) return -1;
    for (int i = 0; i < 10; ++i) out[i] = 0;
    for (int w = 0; w < 4096; ++w)
        for (int i = 0; i < 10; ++i) out[i] += host[w * 10 + i];
    if (reset) { memset(host, 0, sizeof host); if (hipMemcpyToSymbol(HIP_SYMBOL(g_pair_prof), host, sizeof host) != hipSuccess) return -1; }
    return 0;
}
#else
#define RFD_STAMP(i) do { } while (0)
#endif
// NW (round 4): waves per workgroup.  8 = the form above (128-pixel tiles, one workgroup per CU, the weight stream a whole chunk
// ahead in an S + 1 slot ring, one drain per chunk).  4 = HALF workgroups: 64-pixel tiles, a 2-slot weight ring with the next
// step issued one step ahead and drained at the top of every step (conv_igemm's scheme: the L2-served 16-KiB step lands in
// ~170 ns, tools/ring_fill_bench.hip), 45 KiB of LDS -- so TWO workgroups share a CU and run out of phase: one's chunk epilogue
// (VALU, 27 % of a wave's lifetime with the matrix pipe idle) and barrier stalls fall under the other's MFMA steps.  Each wave
// still owns 16 pixels with the same arithmetic in the same order: bit-identical.
// PX (round 4): 16-pixel groups per wave.  2 = a wave owns 32 pixels: every weight fragment read from LDS feeds TWO MFMAs (the
// 16-pixel forms read one fragment per MFMA -- 8 MiB of LDS reads per 128 pixels, as many LDS cycles as the tile has MFMA cycles);
// 4 waves x 32 pixels, one wave per SIMD with up to 512 registers, the deep ring.
// HALF1 (round 4): conv1 has 64 outputs (stage 1's units: 256 -> 64).  W1's 64 rows fill the first half of a slot (the other
// half is never staged nor read), conv1 runs 4 row blocks per step instead of 8, t1's row pitch is 64.
template <int NK, int N1B, bool ACT_OUT, int NK2 = 0, int NCR = 0, int NW = 8, int PX = 1, bool HALF1 = false>
__global__ void __launch_bounds__(NW * 64, PX == 2 ? 1 : 2) pw_pair_kernel(const ConvParams p) // PX 1: 2 waves per SIMD (256 registers: two half workgroups fit a CU)
{
    RFD_CLOCK(3);
#ifdef RFD_PAIR_STAMPS
    unsigned long long prof[10] = {}, tlast = __builtin_readcyclecounter();
    const unsigned long long tstart = tlast;
#endif
    constexpr bool RESIDENT = NCR > 0;
    constexpr bool SHORT = NW == 4 && PX == 1; // 2-slot ring, one step of lead, a drain per step
    static_assert((NW == 8 && PX == 1) || (NW == 4 && !RESIDENT && (PX == 1 || PX == 2)), "8 waves x 16 px, 4 x 16 (short ring) or 4 x 32");
    constexpr int BM = 16 * NW * PX, NT = 64 * NW, PQ = 16 / NW; // pixels per tile, threads, weight pieces per wave and step
    static_assert(!HALF1 || (N1B == 1 && NW == 8), "64-output conv1: one (half) row block, 8 waves");
    constexpr int NKT = NK + NK2, S = NKT + 2 * N1B, WSLOTS = RESIDENT ? NCR * S : (SHORT ? 2 : S + 1), N1 = HALF1 ? 64 : 128 * N1B;
    constexpr int RB1 = HALF1 ? 4 : 8; // 16-row blocks of a W1 slot-step that hold filter rows
    constexpr bool HAS_RES = NK2 == 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Ws = reinterpret_cast<bf16_t *>(smem);                    // [WSLOTS][128*64]
    float *Tab = reinterpret_cast<float *>(Ws + WSLOTS * 128 * 64);  // bias3 [N] | scale [N] | shift [N] | bias1 [N1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.B * p.Ho * p.Wo, K = p.Cin, N = p.Cout, KT = p.Cin + (NK2 ? p.Cin2 : 0);
    const int NC = N >> 7;
    const int tiles_m = (M + BM - 1) / BM;
    const int lr = lane >> 3, chunk = (lane & 7) ^ lr, frow = lane & 15, fq = lane >> 4;
    if ((int)blockIdx.x >= tiles_m) return;

    for (int round = 0; round < (N + NT - 1) / NT; ++round) { // scalar trip count, predicated body
        const int c = tid + round * NT;
        if (c < N) {
            Tab[c] = NK2 ? p.bias[c] + p.bias2[c] : p.bias[c]; // conv_epilogue adds the two biases first, then the accumulator
            Tab[N + c] = p.scale2[c];
            Tab[2 * N + c] = p.shift2[c];
        }
        if (c < N1) Tab[3 * N + c] = p.bias1[c];
    }
    __syncthreads();

    const uint32_t w1_delta = (uint32_t)((const char *)p.w1 - (const char *)p.w); // one descriptor over both filter banks (see pw_b2b_kernel)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w), 0, w1_delta + (uint32_t)((size_t)N1 * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(ACT_OUT ? p.y2 : p.y, 0, (uint32_t)((size_t)M * N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt1 = __builtin_amdgcn_make_buffer_rsrc(p.t1, 0, (uint32_t)((size_t)M * N1 * 2), 0x00020000);
    const u32x4 rres = make_srd(p.res, (uint32_t)((size_t)M * N * 2));
    const u32x4 rxs = make_srd(p.x, (uint32_t)((size_t)M * K * 2));
    const u32x4 rxs2 = make_srd(NK2 ? p.x2 : p.x, (uint32_t)(NK2 ? (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2 : 0));
    uint32_t woff[PQ], woff1[PQ];
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int rho = (wave + NW * q) * 8 + lr;
        const int rw_ = rho & 63, i_ = rw_ >> 4, fq_ = (rw_ >> 2) & 3, r_ = rw_ & 3;
        const int chn = (rho - rw_) + (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_;
        woff[q] = (uint32_t)(((size_t)chn * KT + chunk * 8) * 2);
        woff1[q] = (uint32_t)(((size_t)chn * N + chunk * 8) * 2);
    }
    // unified weight stream, per chunk: W3(c, 0..NK-1), then for each 64-channel half k of the chunk and each 128-row block r of
    // W1: W1(r, k); position of the NEXT step to issue
    int wi_nc = 0, wi_s = 0, wi_slot = 0;
    auto issue_w = [&]() {
        bf16_t *dst = Ws + wi_slot * 128 * 64;
        const bool w3 = wi_s < NKT; // wave-uniform
        const int s1 = wi_s - NKT, k1 = s1 / N1B, r1 = s1 - k1 * N1B;
        const uint32_t so = w3 ? (uint32_t)((((size_t)wi_nc * 128) * KT + (wi_s << 6)) * 2)
                               : w1_delta + (uint32_t)((((size_t)r1 * 128) * N + (wi_nc << 7) + (k1 << 6)) * 2);
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            if (HALF1 && !w3 && q * NW >= 8) continue; // slot rows 64 .. 127 of a 64-row W1: nothing there (NW = 8: pieces q >= 1)
            blds16(rw, w3 ? woff[q] : woff1[q], (uint32_t)__builtin_amdgcn_readfirstlane(so), dst + (wave + NW * q) * 512);
        }
        if (++wi_s == S) { wi_s = 0; if (++wi_nc == NC) wi_nc = 0; }
        wi_slot = wi_slot + 1 == WSLOTS ? 0 : wi_slot + 1;
    };
    // this lane: pixel row wave*16 + frow of the tile; per chunk its four 8-channel groups h*32 + fq*8 (h = 0..3)
    // (pixel group g of the wave: tile rows (wave * PX + g) * 16 + frow)
    auto issue_res = [&](u32x4 (&r)[PX][4], int mt, int nc) {
        if (!HAS_RES) return;
#pragma unroll
        for (int g = 0; g < PX; ++g) {
            const int m = mt * BM + (wave * PX + g) * 16 + frow;
            const bool ok = mt < tiles_m && m < M;
#pragma unroll
            for (int h = 0; h < 4; ++h)
                asm_buffer_load_b128(r[g][h], ok ? (uint32_t)(((size_t)m * N + (nc << 7) + h * 32 + fq * 8) * 2) : kOob, rres);
        }
    };
    u32x4 xq[PX][NKT * 2]; // the activation tile of this wave as B fragments: K slice q = channels q*32 + fq*8 .. +7 of pixel `frow`
    auto issue_x = [&](int mt) {
#pragma unroll
        for (int g = 0; g < PX; ++g) {
            const int m = mt * BM + (wave * PX + g) * 16 + frow;
            const bool ok = mt < tiles_m && m < M;
#pragma unroll
            for (int q = 0; q < NK * 2; ++q) asm_buffer_load_b128(xq[g][q], ok ? (uint32_t)(((size_t)m * K + q * 32 + fq * 8) * 2) : kOob, rxs);
            if (NK2) { // the shortcut's source pixel: (b, stride2 * ho, stride2 * wo) of x2
                const int HoWo = p.Ho * p.Wo, mm = ok ? m : 0, b = mm / HoWo, rem = mm - b * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const uint32_t base = (uint32_t)(((((size_t)b * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.Cin2 + fq * 8) * 2);
#pragma unroll
                for (int q = 0; q < NK2 * 2; ++q) asm_buffer_load_b128(xq[g][NK * 2 + q], ok ? base + (uint32_t)(q * 64) : kOob, rxs2);
            }
        }
    };

    u32x4 resA[PX][4] = {}, resB[PX][4] = {};
    if (RESIDENT) { // the whole of both filter banks, once: NCR chunks x S steps fill the NCR * S slots in consumption order
        for (int i = 0; i < WSLOTS; ++i) issue_w();
        wait_vmcnt<0>();
        __syncthreads();
    }
    issue_x(blockIdx.x);
    issue_res(resA, blockIdx.x, 0);
    if (!RESIDENT) {
#pragma unroll
        for (int i = 0; i < (SHORT ? 1 : S); ++i) issue_w(); // the first chunk's steps (SHORT: the first step)
    }
    int cslot = 0;
    const int arow = frow * 64; // A fragment of row block i: row i*16 + frow; 16-byte slot (kk*4 + fq) ^ (row & 7), row & 7 = frow & 7

    for (int mt = blockIdx.x; mt < tiles_m; mt += gridDim.x) {
        f32x4 acc1[PX][8 * N1B];
#pragma unroll
        for (int g = 0; g < PX; ++g)
#pragma unroll
            for (int i = 0; i < 8 * N1B; ++i) acc1[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the tile's activation fragments (requested during the previous tile's last chunk, or before the loop) and everything older
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        // ... and xq[] is only defined from here on: tied to the wait as cur[] is below, so that no copy of a fragment register
        // hipcc might make (a phi across the back edge, a spill-free re-allocation) can be taken before the loads have landed
        // (volatile asm statements keep their order; round-3 advisor finding)
#pragma unroll
        for (int g = 0; g < PX; ++g)
#pragma unroll
            for (int q = 0; q < NKT * 2; ++q) asm volatile("" : "+v"(xq[g][q]));
        auto do_chunk = [&](int nc, u32x4 (&cur)[PX][4], u32x4 (&nxt)[PX][4], bool last_chunk) __attribute__((always_inline)) {
            f32x4 acc[PX][8];
#pragma unroll
            for (int g = 0; g < PX; ++g)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) { // conv3 (+ shortcut) steps: all 128 rows of the slot against this wave's 16 pixels
                RFD_STAMP(2);
                if (!RESIDENT) {
                    if (SHORT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this step's weights (issued one step ago) and everything older
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    RFD_STAMP(0);
                    issue_w();
                    RFD_STAMP(1);
                }
                const bf16_t *ws = Ws + cslot * 128 * 64 + arow;
                cslot = cslot + 1 == WSLOTS ? 0 : cslot + 1;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int so = ((kk * 4 + fq) ^ (frow & 7)) << 3;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(ws + i * 1024 + so);
#pragma unroll
                        for (int g = 0; g < PX; ++g) // one fragment read, PX MFMAs
                            acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, xq[g][kt * 2 + kk]), acc[g][i], 0, 0, 0);
                    }
                }
            }
            RFD_STAMP(2);
            // the next tile's activation fragments may be requested as soon as this tile's last conv3 step has read them
            if (last_chunk) issue_x(mt + (int)gridDim.x);
            // ---- the one drain of the chunk: residual of this chunk, W1 steps of this chunk, stores of the previous one ----
            if (HAS_RES) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0][0]), "+v"(cur[0][1]), "+v"(cur[0][2]), "+v"(cur[0][3]) : : "memory");
                if (PX == 2) asm volatile("" : "+v"(cur[PX - 1][0]), "+v"(cur[PX - 1][1]), "+v"(cur[PX - 1][2]), "+v"(cur[PX - 1][3])); // behind the wait (volatile asm keeps its order)
            } else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            __builtin_amdgcn_sched_barrier(0);
            RFD_STAMP(3);
            if (!last_chunk) issue_res(nxt, mt, nc + 1);
            else issue_res(nxt, mt + (int)gridDim.x, 0);
            u32x4 actq[PX][4]; // conv1's B fragments: the activated chunk, K slice h = channels h*32 + fq*8 .. +7 of this pixel
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int n = (nc << 7) + h * 32 + fq * 8;
                float bias[8], sc[8], sh[8];
                lds_table_read8x3(Tab + n, Tab + N + n, Tab + 2 * N + n, bias, sc, sh);
#pragma unroll
              for (int g = 0; g < PX; ++g) {
                const int m = mt * BM + (wave * PX + g) * 16 + frow;
                const uint32_t off = m < M ? (uint32_t)(((size_t)m * N + n) * 2) : kOob;
                const u32x4 rv = cur[g][h];
                float v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = acc[g][2 * h][k] + bias[k];
                    v[4 + k] = acc[g][2 * h + 1][k] + bias[4 + k];
                }
                if (HAS_RES) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] += bf16_bits_to_f32(rv[k] & 0xffffu);
                        v[2 * k + 1] += bf16_bits_to_f32(rv[k] >> 16);
                    }
                }
                uint2 alo, ahi;
                if (ACT_OUT) {
                    float a[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
                    alo = pack_bf16x4(a[0], a[1], a[2], a[3]); ahi = pack_bf16x4(a[4], a[5], a[6], a[7]);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{alo.x, alo.y, ahi.x, ahi.y}, ry, off, 0, 0);
                } else {
                    const uint2 lo = pack_bf16x4(v[0], v[1], v[2], v[3]), hi = pack_bf16x4(v[4], v[5], v[6], v[7]);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry, off, 0, 0);
                    const uint32_t rb[4] = {lo.x, lo.y, hi.x, hi.y};
                    float a[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (HALF1) { // stage 1's kernels (conv_b2b_s1_*) round the product, then the sum: the same here, bit for bit
                            a[2 * k] = fmaxf(bf16_bits_to_f32(rb[k] & 0xffffu) * sc[2 * k] + sh[2 * k], 0.f);
                            a[2 * k + 1] = fmaxf(bf16_bits_to_f32(rb[k] >> 16) * sc[2 * k + 1] + sh[2 * k + 1], 0.f);
                        } else {     // beyond stage 1 the pair's second conv is launch_conv's input affine: one fused multiply-add
                            a[2 * k] = fmaxf(__builtin_fmaf(bf16_bits_to_f32(rb[k] & 0xffffu), sc[2 * k], sh[2 * k]), 0.f);
                            a[2 * k + 1] = fmaxf(__builtin_fmaf(bf16_bits_to_f32(rb[k] >> 16), sc[2 * k + 1], sh[2 * k + 1]), 0.f);
                        }
                    }
                    alo = pack_bf16x4(a[0], a[1], a[2], a[3]); ahi = pack_bf16x4(a[4], a[5], a[6], a[7]);
                }
                actq[g][h] = u32x4{alo.x, alo.y, ahi.x, ahi.y};
              }
            }
#pragma unroll
            for (int k1 = 0; k1 < 2; ++k1)       // conv1: the chunk's 64-channel half k1 ...
#pragma unroll
                for (int r1 = 0; r1 < N1B; ++r1) { // ... against the 128-row block r1 of W1 (one slot-step)
                    RFD_STAMP(k1 == 0 && r1 == 0 ? 4 : 7);
                    if (!RESIDENT) {
                        if (SHORT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        RFD_STAMP(5);
                        issue_w();
                        RFD_STAMP(6);
                    }
                    const bf16_t *ws = Ws + cslot * 128 * 64 + arow;
                    cslot = cslot + 1 == WSLOTS ? 0 : cslot + 1;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int so = ((kk * 4 + fq) ^ (frow & 7)) << 3;
#pragma unroll
                        for (int i = 0; i < RB1; ++i) {
                            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(ws + i * 1024 + so);
#pragma unroll
                            for (int g = 0; g < PX; ++g)
                                acc1[g][r1 * 8 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, actq[g][k1 * 2 + kk]), acc1[g][r1 * 8 + i], 0, 0, 0);
                        }
                    }
                }
        };
        for (int nc = 0; nc < NC; nc += 2) {
            do_chunk(nc, resA, resB, false);
            RFD_STAMP(7);
            do_chunk(nc + 1, resB, resA, nc + 2 >= NC);
            RFD_STAMP(7);
        }
        // ---- conv1 epilogue: t1 = relu(acc1 + bias1): per 128-row block four 8-channel groups of this pixel ----
#pragma unroll
        for (int r1 = 0; r1 < N1B; ++r1)
#pragma unroll
            for (int h = 0; h < RB1 / 2; ++h) {
                const int n = r1 * 128 + h * 32 + fq * 8;
                float b1[8];
                lds_table_read8(Tab + 3 * N + n, b1);
#pragma unroll
                for (int g = 0; g < PX; ++g) {
                    const int m = mt * BM + (wave * PX + g) * 16 + frow;
                    float o[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        o[k] = fmaxf(acc1[g][r1 * 8 + 2 * h][k] + b1[k], 0.f);
                        o[4 + k] = fmaxf(acc1[g][r1 * 8 + 2 * h + 1][k] + b1[4 + k], 0.f);
                    }
                    const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, rt1, m < M ? (uint32_t)(((size_t)m * N1 + n) * 2) : kOob, 0, 0);
                }
            }
    }
#ifdef RFD_PAIR_STAMPS
    RFD_STAMP(4); // the last tile's conv1 epilogue counts as epilogue
#endif
    wait_vmcnt<0>(); // weight steps, residual and activation loads issued beyond the end
#ifdef RFD_PAIR_STAMPS
    RFD_STAMP(8);
    prof[9] = __builtin_readcyclecounter() - tstart;
    if (lane == 0 && blockIdx.x < 512)
        for (int i = 0; i < 10; ++i) g_pair_prof[(blockIdx.x * 8 + wave) * 10 + i] = prof[i];
#endif
}

// CU count of the current device, queried once per device (every persistent launcher sizes its "even share, no tail" grid
// with it; a partitioned or smaller device simply gets a smaller grid)
static int device_cus()
{
    static std::atomic<int> cached[DynLdsOnce::kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DynLdsOnce::kMaxDevices) return 256;
    int n = cached[dev].load(std::memory_order_acquire);
    if (n > 0) return n;
    hipDeviceProp_t pr;
    n = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    cached[dev].store(n, std::memory_order_release);
    return n;
}

// CUs a persistent kernel spreads over when another chain of the same pass runs beside it (RFD_PERSIST_CUS: A/B knob;
// default: all of them)
static int persistent_cus(int co_running, bool hbm_bound = false)
{
    static const int env = [] { const char *e = getenv("RFD_PERSIST_CUS"); return e ? atoi(e) : 0; }();
    static const int env_hbm = [] { const char *e = getenv("RFD_PERSIST_CUS_HBM"); return e ? atoi(e) : 0; }(); // HBM-bound kernels only
    const int ncu = device_cus();
    if (co_running && hbm_bound && env_hbm > 0) return std::min(env_hbm, ncu);
    return co_running && env > 0 ? std::min(env, ncu) : ncu;
}

// ---- the one launch path of every persistent kernel ----
// A persistent workgroup must own its CU's LDS (rule 2): whatever the kernel needs, the launch asks for
// the whole 160 KiB, so no workgroup of another kernel can ever share the CU.  kPersistentKernels is the list the CPU build test
// walks (tests/test_build_cpu.py, through rfd_debug_persistent_kernel): every 8-wave LDS-DMA kernel of the code object is either
// in it or named there as one-tile-per-workgroup.  RFD_PERSIST_LDS_EXACT=1 (diagnostic only, tools/split_diag.py) requests the
// real need instead -- the configuration that gave nondeterministic images in round 2.
constexpr size_t kPersistentLds = 160 * 1024;
static const char *const kPersistentKernels[] = {"pw_stream_kernel", "conv3x3_c64_kernel", "conv3x3_halo_kernel", "pw_gemm_kernel",
                                                 "pw_wide_kernel", "conv_b2b_s1_persistent_kernel", "conv_b2b_s1_persistent_k128_kernel",
                                                 "pw_b2b_kernel", "pw_pair_kernel"};
int persistent_kernel_table(int i, const char **name, size_t *lds_bytes)
{
    const int n = (int)(sizeof(kPersistentKernels) / sizeof(kPersistentKernels[0]));
    if (i < 0 || i >= n) return n;
    if (name) *name = kPersistentKernels[i];
    if (lds_bytes) *lds_bytes = kPersistentLds;
    return n;
}
template <auto Kern, typename... A> static int launch_persistent(int grid, size_t lds_need, hipStream_t s, A... args)
{
    if (launch_note().dry) return RFD_OK; // the caller has recorded the kernel's name (note_launch)
    static const bool exact = [] { const char *e = getenv("RFD_PERSIST_LDS_EXACT"); return e && atoi(e) != 0; }();
    if (lds_need > kPersistentLds) { set_error("persistent kernel: %zu bytes of LDS needed", lds_need); return RFD_ERR_CAPACITY; }
    static DynLdsOnce once;
    RFD_TRY(once.ensure(reinterpret_cast<const void *>(Kern), (int)kPersistentLds));
    hipLaunchKernelGGL(Kern, dim3(grid), dim3(512), exact ? lds_need : kPersistentLds, s, args...);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

template <int NK, bool HAS_Y, bool HAS_Y2> static int launch_pw_stream(const ConvParams &p, hipStream_t s)
{
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = ceil_div(M, 128);
    const int ncu = persistent_cus(p.co_running, true);
    // one persistent workgroup per CU; tiles are dealt round-robin, so an even share per workgroup means no tail
    const int per = ceil_div(tiles_m, ncu);
    const int grid = ceil_div(tiles_m, per);
    // The kernel needs (2 NK + 1) x 16 KiB + 12 B per output channel (86 KiB at K = 128, 156 KiB at K = 256); launch_persistent
    // asks for the CU's whole LDS.  At 86 KiB the K = 128 variant co-resided with the other chain's 3x3 workgroups in the split
    // mode, and that configuration produced nondeterministic outputs in round 2 (2-3 images of a 16-image part wrong in 5 of 8
    // passes; never with the 156 KiB variant, never once the request was padded: DESIGN.md section 5).  Round 3 found a
    // write-after-read race on the weight ring in this kernel's ISA (bare s_barrier with ds_reads in flight, now
    // `s_waitcnt lgkmcnt(0)` + barrier) that the padding may only have masked; the padding stays as the rule either way.
    const size_t lds_need = (size_t)(NK * 128 + (NK + 1) * 128) * 64 * sizeof(bf16_t) + (size_t)3 * p.Cout * sizeof(float);
    note_launch("pw_stream_kernel<%d, %s, %s>", NK, HAS_Y ? "true" : "false", HAS_Y2 ? "true" : "false");
    return launch_persistent<pw_stream_kernel<NK, HAS_Y, HAS_Y2>>(grid, lds_need, s, p);
}
template <int NK, bool ACT_OUT> static int launch_pw_b2b(const ConvParams &p, hipStream_t s)
{
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = ceil_div(M, 128);
    const int ncu = persistent_cus(p.co_running, true);
    const int per = ceil_div(tiles_m, ncu);
    const int grid = ceil_div(tiles_m, per);
    const size_t lds_need = (size_t)(NK + 2 + NK + 3) * 128 * 64 * sizeof(bf16_t) + (size_t)(3 * p.Cout + 128) * sizeof(float);
    note_launch("pw_b2b_kernel<%d, %s>", NK, ACT_OUT ? "true" : "false");
    return launch_persistent<pw_b2b_kernel<NK, ACT_OUT>>(grid, lds_need, s, p);
}
// half-workgroup form of pw_pair_kernel (NW = 4): two workgroups per CU, 64-pixel tiles, short weight ring
template <int NK, int N1B, bool ACT_OUT, int NK2 = 0> static int launch_pw_pair_half(const ConvParams &p, hipStream_t s)
{
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = ceil_div(M, 64);
    const int slots = 2 * persistent_cus(p.co_running, true);
    const int per = ceil_div(tiles_m, slots);
    const int grid = ceil_div(tiles_m, per);
    const size_t lds_need = (size_t)2 * 128 * 64 * sizeof(bf16_t) + (size_t)(3 * p.Cout + 128 * N1B) * sizeof(float);
    // each workgroup asks for HALF the CU's LDS, so that exactly two co-reside (and nothing else beside them)
    constexpr size_t kHalfLds = 80 * 1024;
    if (lds_need > kHalfLds) { set_error("pw_pair (half workgroups): %zu bytes of LDS needed", lds_need); return RFD_ERR_CAPACITY; }
    note_launch("pw_pair_kernel<%d, %d, %s, %d, 0, 4, 1, false>", NK, N1B, ACT_OUT ? "true" : "false", NK2); // the name rocprofv3 reports: every template argument
    if (launch_note().dry) return RFD_OK;
    auto kern = pw_pair_kernel<NK, N1B, ACT_OUT, NK2, 0, 4>;
    static DynLdsOnce once;
    RFD_TRY(once.ensure(reinterpret_cast<const void *>(kern), (int)kHalfLds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kHalfLds, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}
// 32-pixel-per-wave form of pw_pair_kernel (NW = 4, PX = 2): 128-pixel tiles, one workgroup of 4 waves per CU, the deep ring
template <int NK, int N1B, bool ACT_OUT, int NK2 = 0> static int launch_pw_pair_px2(const ConvParams &p, hipStream_t s)
{
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = ceil_div(M, 128);
    const int ncu = persistent_cus(p.co_running, true);
    const int per = ceil_div(tiles_m, ncu);
    const int grid = ceil_div(tiles_m, per);
    const size_t lds_need = (size_t)(NK + NK2 + 2 * N1B + 1) * 128 * 64 * sizeof(bf16_t) + (size_t)(3 * p.Cout + 128 * N1B) * sizeof(float);
    if (lds_need > kPersistentLds) { set_error("pw_pair (32 px per wave): %zu bytes of LDS needed", lds_need); return RFD_ERR_CAPACITY; }
    note_launch("pw_pair_kernel<%d, %d, %s, %d, 0, 4, 2, false>", NK, N1B, ACT_OUT ? "true" : "false", NK2);
    if (launch_note().dry) return RFD_OK;
    auto kern = pw_pair_kernel<NK, N1B, ACT_OUT, NK2, 0, 4, 2>;
    static DynLdsOnce once;
    RFD_TRY(once.ensure(reinterpret_cast<const void *>(kern), (int)kPersistentLds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kPersistentLds, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}
template <int NK, int N1B, bool ACT_OUT, int NK2 = 0, int NCR = 0, bool HALF1 = false> static int launch_pw_pair(const ConvParams &p, hipStream_t s)
{
    // RFD_PAIR_HALF: which pairs run as half workgroups (bit mask: 1 stage 3's middle units <4,2>, 2 the 2 -> 3 boundary <2,2,true>,
    // 4 stage 2's first unit <2,1,false,4>, 8 stage 2's middle units); A/B knob, bit-identical either way
    static const int half_env = [] { const char *e = getenv("RFD_PAIR_HALF"); return e ? atoi(e) : 0; }();
    if (NCR == 0 && !HALF1) {
        const int bit = (NK == 4 && N1B == 2) ? 1 : (NK == 2 && N1B == 2) ? 2 : (NK == 2 && NK2 == 4) ? 4 : (NK == 2 && N1B == 1 && NK2 == 0) ? 8 : 0;
        if (half_env & bit) return launch_pw_pair_half<NK, N1B, ACT_OUT, NK2>(p, s);
        // RFD_PAIR_PX2: the same bit mask for the 32-pixel-per-wave form
        static const int px2_env = [] { const char *e = getenv("RFD_PAIR_PX2"); return e ? atoi(e) : 0; }();
        if (px2_env & bit) return launch_pw_pair_px2<NK, N1B, ACT_OUT, NK2>(p, s);
    }
    const int M = p.B * p.Ho * p.Wo;
    const int tiles_m = ceil_div(M, 128);
    const int ncu = persistent_cus(p.co_running, true);
    const int per = ceil_div(tiles_m, ncu);
    const int grid = ceil_div(tiles_m, per);
    const size_t slots = NCR ? (size_t)NCR * (NK + NK2 + 2 * N1B) : (size_t)(NK + NK2 + 2 * N1B + 1);
    const size_t lds_need = slots * 128 * 64 * sizeof(bf16_t) + (size_t)(3 * p.Cout + 128 * N1B) * sizeof(float);
    if (NCR && p.Cout != NCR * 128) { set_error("pw_pair: resident form instantiated for %d output channels", NCR * 128); return RFD_ERR_INVALID_ARG; }
    if (p.n1 != (HALF1 ? 64 : 128 * N1B)) { set_error("pw_pair: instantiated for n1 = %d, got %d", HALF1 ? 64 : 128 * N1B, p.n1); return RFD_ERR_INVALID_ARG; }
    note_launch("pw_pair_kernel<%d, %d, %s, %d, %d, 8, 1, %s>", NK, N1B, ACT_OUT ? "true" : "false", NK2, NCR, HALF1 ? "true" : "false");
    return launch_persistent<pw_pair_kernel<NK, N1B, ACT_OUT, NK2, NCR, 8, 1, HALF1>>(grid, lds_need, s, p);
}
template <int NK> static int launch_pw_stream_nk(const ConvParams &p, hipStream_t s)
{

    if (p.y && p.y2) return launch_pw_stream<NK, true, true>(p, s);
    if (p.y) return launch_pw_stream<NK, true, false>(p, s);
    return launch_pw_stream<NK, false, true>(p, s);
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolutions with the three kx taps sharing ONE staged activation tile.
//
// The generic kernel above is LDS-DMA *issue* bound (8 one-KiB pieces per 32 MFMAs per wave; dropping the
// activation DMA in a timing-only build made 3x3 layers 23-30 % faster).  For a 3x3 stride-1 conv the taps
// (ky, kx = 0,1,2) of one 64-channel chunk read the same pixels shifted by ONE pixel, i.e. by one 128-byte
// row of the flattened-pixel LDS tile.  So per (ky, chunk) one EXTENDED tile -- pixels m0-1 .. m0+158, 20
// pieces -- is staged, and the three kx K-steps read their B fragments at row offsets 0 / 1 / 2; a fragment
// whose pixel sits in image column 0 (kx = 0) or W-1 (kx = 2) would wrap into the neighbouring image row
// and is zeroed in registers instead.  Activation pieces per three K-steps: 12 -> 5 per wave; the weight
// tile streams as before.  K order: (chunk, ky, kx) -- only the f32 summation order differs from the generic kernel's
// (ky, kx, chunk); it is the order of conv3x3_halo_kernel below, so the two give bit-identical results.
// LDS: 2 x 20 KiB activation slots + 2 weight slots: 72 KiB at BN = 128 -> two workgroups per CU.
// ------------------------------------------------------------------------------------------------
template <int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(WAVES_M *WAVES_N * 64, 2) conv3x3_kx_kernel(const ConvParams p) // two waves per SIMD: <= 256 registers, so that two 4-wave workgroups share a CU (a guard in stage_x once cost 20 registers and the co-residency: 67 -> 108 us)
{
    RFD_CLOCK(4);
    constexpr int BM = 128, XE = 160; // extended tile rows (20 pieces)
    constexpr int NW = WAVES_M * WAVES_N;
    static_assert(NW == 4 || NW == 8, "4 waves (64 x 64 wave tiles) or 8 (32 x 64: two waves per SIMD from ONE workgroup, for grids that give a CU a single workgroup)");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int XPE = (XE / 8 + NW - 1) / NW, WP = (BN / 8 + NW - 1) / NW; // 5 (8 waves: 3, the last for waves 0-3 only) activation pieces per wave and (ky, chunk)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem); // [2][XE*64]
    bf16_t *Ws = Xs + 2 * XE * 64;                  // [2][BN*64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int HW = p.H * p.W;
    const int M = p.B * HW;
    const int K = 9 * p.Cin;
    const int kc_n = p.Cin >> 6, ngroups = 3 * kc_n;
    const int tiles_n = p.Cout / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

    const int lr = lane >> 3, chunk = (lane & 7) ^ lr;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)p.B * HW * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.w), 0, (uint32_t)((size_t)p.Cout * K * 2), 0x00020000);
    // extended-tile row rho = piece*8 + lr holds pixel q = m0 - 1 + rho at input row y(q) + ky - 1
    uint32_t xoff[XPE];
    int y0[XPE];
#pragma unroll
    for (int q = 0; q < XPE; ++q) {
        const int pix = m0 - 1 + (wave + NW * q) * 8 + lr;
        y0[q] = -(1 << 28);
        xoff[q] = 0;
        if (pix >= 0 && pix < M) {
            const int b = pix / HW, rem = pix - b * HW;
            const int y = rem / p.W, x = rem - y * p.W;
            y0[q] = y - 1;
            xoff[q] = (uint32_t)(((((long long)b * p.H + y - 1) * p.W + x) * p.ldx + p.x_coff + chunk * 8) * 2);
        }
    }
    uint32_t woff[WP];
#pragma unroll
    for (int q = 0; q < WP; ++q) {
        const int piece = wave + NW * q;
        const int rho = (piece < BN / 8 ? piece * 8 + lr : 0);
        const int rw_ = rho % WN, i_ = rw_ >> 4, fq_ = (rw_ >> 2) & 3, r_ = rw_ & 3;
        const int chn = (rho - rw_) + (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_;
        woff[q] = (uint32_t)(((size_t)(n0 + chn) * K + chunk * 8) * 2);
    }
    auto stage_x = [&](int slot, int ky, int kc) {
        const uint32_t rowoff = (uint32_t)(ky * p.W * p.ldx * 2);
#pragma unroll
        for (int q = 0; q < XPE; ++q) {
            if (XPE * NW > XE / 8 && wave + NW * q >= XE / 8) continue; // 8 waves: the last round is for waves 0-3 only (wave-uniform)
            const bool ok = (unsigned)(y0[q] + ky) < (unsigned)p.H;
            blds16(rx, ok ? xoff[q] + rowoff : kOob, (uint32_t)(kc << 7), Xs + slot * XE * 64 + (wave + NW * q) * 512);
        }
    };
    auto stage_w = [&](int slot, int ky, int kc, int kx) {
        const uint32_t col = (uint32_t)((((ky * 3 + kx) * p.Cin) + (kc << 6)) * 2);
#pragma unroll
        for (int q = 0; q < WP; ++q) {
            const int piece = wave + NW * q;
            if (piece < BN / 8) blds16(rw, woff[q], col, Ws + slot * BN * 64 + piece * 512);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    constexpr int TH = TN / 2;
    uint4 resv[TM][TH];
    conv_prefetch_residual<TM, TH, WM, WN>(p, resv, m0, n0, wm, wn, frow, fq, M, HW);
    // fragment (j) is pixel m0 + wm*WM + j*16 + frow: column 0 kills the kx = 0 tap, column W-1 the kx = 2 tap
    bool col_first[TM], col_last[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int x = (m0 + wm * WM + j * 16 + frow) % p.W;
        col_first[j] = x == 0;
        col_last[j] = x == p.W - 1;
    }

    // queue order: X(group 0), W(step 0)
    stage_x(0, 0, 0);
    stage_w(0, 0, 0, 0);
    int ky = 0, kc = 0; // current group
    int xslot = 0, wslot = 0;
    for (int g = 0; g < ngroups; ++g) {
        int nky = ky + 1, nkc = kc; // next group: chunk-major, the order conv3x3_halo_kernel is bound to (bit-identical results)
        if (nky == 3) { nky = 0; ++nkc; }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // step (g, kx) needs W(this step) and X(g); X(g+1) was requested in the kx = 0 step and lands with this drain too
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* no LDS read in flight at a barrier that frees a ring slot for DMA (tools/isa_check.py) */
            if (kx < 2) stage_w(wslot ^ 1, ky, kc, kx + 1);
            else if (g + 1 < ngroups) stage_w(wslot ^ 1, nky, nkc, 0);
            if (kx == 0 && g + 1 < ngroups) stage_x(xslot ^ 1, nky, nkc);
            const bf16_t *xs = Xs + xslot * XE * 64 + (wm * WM + kx) * 64; // tap kx: one LDS row further
            const bf16_t *ws = Ws + wslot * BN * 64 + (wn * WN) * 64;
            wslot ^= 1;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[TN], bfr[TM];
                const int ch = kk * 4 + fq;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int r = i * 16 + frow;
                    af[i] = *reinterpret_cast<const bf16x8 *>(ws + r * 64 + ((ch ^ (r & 7)) << 3));
                }
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int r = j * 16 + frow;                    // row within the wave's slice
                    const int rho = wm * WM + kx + r;               // LDS row of the extended tile: sets the swizzle
                    bf16x8 v = *reinterpret_cast<const bf16x8 *>(xs + r * 64 + ((ch ^ (rho & 7)) << 3));
                    if ((kx == 0 && col_first[j]) || (kx == 2 && col_last[j])) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    bfr[j] = v;
                }
    #pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
        }
        xslot ^= 1;
        ky = nky; kc = nkc;
    }
    conv_epilogue<TM, TN, WM, WN>(p, acc, resv, m0, n0, wm, wn, frow, fq, M);
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1, 64 -> 64 channels (conv2 of the stage-1 units at 160 x 160, the SSH context convs): the whole
// filter bank is 72 KiB, so a PERSISTENT workgroup keeps it in LDS for its lifetime and walks 16 x 16-pixel output tiles;
// per tile only the 18 x 18-pixel halo tile (41 KiB) is staged -- ONCE for all nine taps -- into a double buffer while the
// previous tile computes.  Through the CU's load path that is 41 KiB in + 32 KiB out per 18.9 MFLOP (257 FLOP/B) where
// the generic 128 x 64 tile moves 221 KiB + 16 KiB per 9.4 MFLOP (40 FLOP/B); the generic kernel ran these layers at
// 13-14 TB/s of L2->LDS traffic, i.e. load-path bound (0.52-0.56 PF/s).
// 8 waves = 4 (pixel rows) x 2 (channels); a wave owns 4 spatial rows x 16 columns x 32 channels; all 18 (tap, kk) steps are
// unrolled with immediate LDS offsets.  The halo image is stored pixel-major (row R = hy * 18 + hx, 128 B per pixel) with the
// chunk swizzle keyed on hx & 7, so the XOR of a tap depends on kx only.  LDS: 72 KiB + 2 x 41 KiB = 154 KiB, 1 workgroup / CU.
// ------------------------------------------------------------------------------------------------
constexpr int kC64T = 16, kC64H = kC64T + 2, kC64HP = 41 /* pieces of 8 rows */;
#ifndef RFD_C64_EXP
#define RFD_C64_EXP 0 // timing experiments (tools/build_variant.sh; results are garbage): 1 no halo DMA after the first tile, 2 no output stores, 3 neither, 4 neither + no MFMAs
#endif
__global__ void __launch_bounds__(512) conv3x3_c64_kernel(const ConvParams p, int tiles_x, int tiles_y)
{
    RFD_CLOCK(5);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Wt = reinterpret_cast<bf16_t *>(smem);       // [9][64][64]
    bf16_t *Xh = Wt + 9 * 64 * 64;                        // [2][kC64HP*8][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lr = lane >> 3, slot = lane & 7, frow = lane & 15, fq = lane >> 4;
    const int ntiles = p.B * tiles_x * tiles_y;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)p.B * p.H * p.W * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w), 0, (uint32_t)(64 * 576 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (uint32_t)((size_t)p.B * p.H * p.W * p.ldy * 2), 0x00020000);
    if ((int)blockIdx.x >= ntiles) return;

    // ---- filter bank -> LDS, once: 72 pieces (tap t, rows 8 q .. 8 q + 7), wave w takes pieces w, w + 8, ...
    //      LDS row rho of a tap holds output channel perm(rho) (8 consecutive channels per lane in the epilogue) ----
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int piece = wave + 8 * i, t = piece >> 3, rho = (piece & 7) * 8 + lr;
        const int rw_ = rho & 31, i_ = rw_ >> 4, fq_ = (rw_ >> 2) & 3, r_ = rw_ & 3;
        const int chn = (rho - rw_) + fq_ * 8 + (i_ & 1) * 4 + r_;
        blds16(rw, (uint32_t)(((size_t)chn * 576 + t * 64 + ((slot ^ lr) << 3)) * 2), 0, Wt + piece * 512);
    }
    // ---- halo staging: piece q of this wave covers image rows R = (wave + 8 q) * 8 + lr of the halo tile ----
    constexpr int HQ = (kC64HP + 7) / 8; // 6 pieces per wave at most
    int hy[HQ], hx[HQ];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const int R = (wave + 8 * q) * 8 + lr;
        hy[q] = R / kC64H;
        hx[q] = R - hy[q] * kC64H;
    }
    auto stage_halo = [&](int tile, int buf) {
        const int b = tile / (tiles_x * tiles_y), rem = tile - b * tiles_x * tiles_y;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int piece = wave + 8 * q;
            if (piece < kC64HP) {
                const int gy = ty * kC64T - 1 + hy[q], gx = tx * kC64T - 1 + hx[q];
                const bool ok = hy[q] < kC64H && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                const uint32_t off = (uint32_t)(((((size_t)b * p.H + gy) * p.W + gx) * p.ldx + p.x_coff + ((slot ^ (hx[q] & 7)) << 3)) * 2);
                blds16(rx, ok ? off : kOob, 0, Xh + buf * (kC64HP * 512) + piece * 512);
            }
        }
    };
    // per-lane read bases: B fragment (j) = pixel (4 wm + j, frow) of the tile -> halo row (4 wm + j + ky) * 18 + frow + kx
    const bf16_t *xbase = Xh + ((wm * 4) * kC64H + frow) * 64;
    const bf16_t *wbase = Wt + (wn * 32 + frow) * 64;
    int xsw[3][2]; // chunk index of (kx, kk) for this lane: (kk * 4 + fq) ^ ((frow + kx) & 7)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xsw[kx][kk] = ((kk * 4 + fq) ^ ((frow + kx) & 7)) << 3;
    const int wsw0 = ((0 * 4 + fq) ^ (frow & 7)) << 3, wsw1 = ((1 * 4 + fq) ^ (frow & 7)) << 3;
    float bias[8];
    {
        const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + wn * 32 + fq * 8), b1 = *reinterpret_cast<const float4 *>(p.bias + wn * 32 + fq * 8 + 4);
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }

    int buf = 0;
    stage_halo(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        // this tile's halo (and, the first time, the filter bank) has landed in every wave (drained below, before the
        // previous tile's stores were issued); the other buffer is free
        if (tile == (int)blockIdx.x) wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* no LDS read in flight at a barrier that frees a ring slot for DMA (tools/isa_check.py) */
#if RFD_C64_EXP == 0 || RFD_C64_EXP == 2
        if (next < ntiles) stage_halo(next, buf ^ 1);
#endif
        const bf16_t *xb = xbase + buf * (kC64HP * 512);
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 af[2], bfr[4];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        af[i] = *reinterpret_cast<const bf16x8 *>(wbase + ((ky * 3 + kx) * 64 + i * 16) * 64 + (kk ? wsw1 : wsw0));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bfr[j] = *reinterpret_cast<const bf16x8 *>(xb + ((j + ky) * kC64H + kx) * 64 + xsw[kx][kk]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
        // the next tile's halo (issued a whole tile ago) and the previous tile's stores: a full drain, BEFORE this tile's
        // stores are issued -- a counted wait behind them would rest on stores never overtaking older LDS-DMA loads
        wait_vmcnt<0>();
        // epilogue: + bias, ReLU, 16-byte stores (8 consecutive channels of one pixel per lane)
        const int b = tile / (tiles_x * tiles_y), rem = tile - b * tiles_x * tiles_y;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int ox = tx * kC64T + frow;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = ty * kC64T + wm * 4 + j;
            float o[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = acc[0][j][k] + bias[k];
                o[4 + k] = acc[1][j][k] + bias[4 + k];
            }
            if (p.relu) {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = fmaxf(o[k], 0.f);
            }
            const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
            const uint32_t yoff = (uint32_t)(((((size_t)b * p.H + oy) * p.W + ox) * p.ldy + p.y_coff + wn * 32 + fq * 8) * 2);
#if RFD_C64_EXP == 0 || RFD_C64_EXP == 1
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry, (oy < p.H && ox < p.W) ? yoff : kOob, 0, 0);
#else
            if (lo.x == 0x12345678u && tile < 0) __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry, yoff, 0, 0); // keeps the values live
#endif
        }
        buf ^= 1;
    }
}

static int launch_conv3x3_c64(const ConvParams &p, hipStream_t s)
{
    const int tiles_x = ceil_div(p.W, kC64T), tiles_y = ceil_div(p.H, kC64T);
    const int ntiles = p.B * tiles_x * tiles_y;
    const int ncu = persistent_cus(p.co_running);
    const int per = ceil_div(ntiles, ncu);
    const int grid = ceil_div(ntiles, per); // even share per persistent workgroup: no tail
    // needs 154 KiB; launch_persistent asks for the whole CU's LDS so that no other kernel's workgroup can ever share the CU
    constexpr size_t lds_need = (size_t)(9 * 64 * 64 + 2 * kC64HP * 512) * sizeof(bf16_t);
    static_assert(lds_need <= kPersistentLds, "LDS");
    note_launch("conv3x3_c64_kernel");
    return launch_persistent<conv3x3_c64_kernel>(grid, lds_need, s, p, tiles_x, tiles_y);
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 with Cin >= 128 (conv2 of the stage-2 / stage-3 units, the FPN aggregation convs): the filter
// bank no longer fits LDS, but the two things that bound the merged-kx kernel above still can be cut -- LDS-DMA pieces
// issued per MFMA and bytes through the CU's load path.  A PERSISTENT workgroup (8 waves, the CU to itself) computes
// 256-pixel x NWG-channel work items:
//   * pixels: a TR x TC tile of one image (16 x 16 at W = 80; 6 full rows at W = 40); its (TR+2) x (TC+2) halo tile of ONE
//     64-channel chunk is staged once for all nine taps (c64's layout: pixel-major, 128 B per pixel, chunk swizzle keyed
//     on the halo column), double-buffered across chunks;
//   * weights: one [NWG][64] tile per (chunk, tap) "unit"; a K step is 32 KiB of them -- one unit at NWG = 256 (TN = 8
//     fragments per wave), two at NWG = 128 (TN = 4) -- in a 2-slot ring: 64 MFMAs per wave between barriers.
// DMA pieces per 64 MFMAs per wave: 4 (weights) + ~0.6 (halo), against 11.4 in the merged-kx kernel; load-path bytes per
// 128 x 128 outputs: 0.26x - 0.5x.
// Waits follow the rule of DESIGN.md section 5 (drain only, and only what was issued at least a step ago), which needs the
// two operand streams in DIFFERENT waves' counters: waves 0-3 issue only weight DMAs (next step's tile, drained at the top
// of every step), waves 4-7 only halo DMAs (next chunk's tile, HBM-served, issued 4-9 steps before their drain).
// Nine steps are unrolled ("superblock": one chunk at TN = 8, two chunks = 18 units at TN = 4, where step 4 straddles the
// chunk boundary and the two halo buffers are bound to chunk parity).
// ------------------------------------------------------------------------------------------------
#ifndef RFD_HALO_EXP
#define RFD_HALO_EXP 0 // timing experiments (tools/build_variant.sh; results are garbage): 1 no weight DMA, 2 no halo DMA, 3 neither, 4 neither + no barrier, 5 weight DMAs issued but never waited for
#endif
template <int TC, int TR, int TN>
__global__ void __launch_bounds__(512) conv3x3_halo_kernel(const ConvParams p, int tiles_x, int tiles_y, int n_items)
{
    RFD_CLOCK(6);
    constexpr int HW2 = TC + 2, HPX = (TR + 2) * HW2, HP = (HPX + 7) / 8; // halo pixels, 8-pixel DMA pieces
    constexpr int HQ = (HP + 3) / 4;                                       // pieces per halo wave
    constexpr int HEL = HP * 512;                                          // elements per halo buffer
    constexpr int NWG = 32 * TN;                                           // output channels per work item
    constexpr int U = TN == 4 ? 2 : 1;                                     // units (taps) per step
    constexpr int WEL = U * NWG * 64;                                      // elements per weight slot
    constexpr int WQ = NWG / 32;                                           // weight pieces per W wave and unit
    constexpr int NPIX = TR * TC;
    static_assert(NPIX <= 256 && NPIX > 192, "four 64-pixel wave rows");
    static_assert((2 * HEL + 2 * WEL) * 2 + 2048 <= 160 * 1024, "LDS");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xh = reinterpret_cast<bf16_t *>(smem); // [2][HEL]
    bf16_t *Ws = Xh + 2 * HEL;                      // [2][WEL]
    float *Tab = reinterpret_cast<float *>(Ws + 2 * WEL); // bias[Cout <= 512]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2; // wn == 0: weight-stream waves, wn == 1: halo-stream waves
    const int lr = lane >> 3, slot = lane & 7, frow = lane & 15, fq = lane >> 4;
    const int K = 9 * p.Cin, KC = p.Cin >> 6;
    const int tiles_n = p.Cout / NWG, tiles_img = tiles_x * tiles_y;
    const int grid = gridDim.x;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)p.B * p.H * p.W * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.w), 0, (uint32_t)((size_t)p.Cout * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (uint32_t)((size_t)p.B * p.H * p.W * p.ldy * 2), 0x00020000);
    int item = xcd_remap(blockIdx.x, grid);
    if (item >= n_items) return;

    for (int c = tid; c < p.Cout; c += 512) Tab[c] = p.bias[c];

    // item -> (n tile, image, tile row, tile column); n fastest: neighbouring workgroups of an XCD share the halo in L2
    auto item_n0 = [&](int it) { return (it % tiles_n) * NWG; };
    // ---- halo stream (waves 4-7): piece q of wave wm covers halo pixels idx = (wm + 4 q) * 8 + lr ----
    auto issue_halo = [&](int it, int chunk, int buf) {
        const int t = it / tiles_n, b = t / tiles_img, rem = t - b * tiles_img;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        int lr_ = lr; // opaque: the per-piece halo coordinates are recomputed here (a few VALU ops once per chunk, in the
        asm volatile("" : "+v"(lr_)); // halo waves only) instead of living in 2 x HQ registers across the MFMA loop
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int piece = wm + 4 * q;
            if (piece < HP) {
                const int idx = piece * 8 + lr_, hy = idx / HW2, hx = idx - hy * HW2;
                const int gy = ty * TR - 1 + hy, gx = tx * TC - 1 + hx;
                const bool ok = idx < HPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                const uint32_t off = (uint32_t)(((((size_t)b * p.H + gy) * p.W + gx) * p.ldx + p.x_coff + ((slot ^ (hx & 7)) << 3)) * 2);
                blds16(rx, ok ? off : kOob, (uint32_t)(chunk << 7), Xh + buf * HEL + piece * 512);
            }
        }
    };
    // ---- weight stream (waves 0-3): LDS row rho of a unit tile holds output channel perm(rho) (8 consecutive channels
    //      per lane in the epilogue); piece q of wave wm = rows (wm + 4 q) * 8 + lr ----
    uint32_t woff[WQ];
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int rho = (wm + 4 * q) * 8 + lr;
        const int wn_ = rho / (16 * TN), rw_ = rho % (16 * TN), i_ = rw_ >> 4, fq_ = (rw_ >> 2) & 3, r_ = rw_ & 3;
        const int chn = wn_ * 16 * TN + (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_;
        woff[q] = (uint32_t)(((size_t)chn * K + ((slot ^ lr) << 3)) * 2);
    }
    auto issue_w = [&](int wslot, int it, int c0, int s) { // step s of the superblock starting at chunk c0 of item it
        const uint32_t rowbase = (uint32_t)((size_t)item_n0(it) * K * 2);
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int u = s * U + k, cs = u / 9, tap = u - cs * 9;
            const uint32_t col = (uint32_t)((tap * p.Cin + ((c0 + cs) << 6)) * 2);
#pragma unroll
            for (int q = 0; q < WQ; ++q)
                blds16(rw, woff[q], rowbase + col, Ws + wslot * WEL + k * (NWG * 64) + (wm + 4 * q) * 512);
        }
    };

    // ---- per-lane fragment addresses ----
    // B fragment (j): pixel slot q = 64 wm + 16 j + frow of the tile -> (r, c); tap (ky, kx) reads halo pixel (r + ky, c + kx)
    int xa[4][3]; // element offset inside a halo buffer of (j, kx) at ky = 0, kk = 0; kk = 1 flips chunk bit 2 (^ 32 elements)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int q = wm * 64 + j * 16 + frow;
        if (q > NPIX - 1) q = NPIX - 1;
        const int r = q / TC, c = q - r * TC;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) xa[j][kx] = (r * HW2 + c + kx) * 64 + ((fq ^ ((c + kx) & 7)) << 3);
    }
    const int wa0 = (wn * 16 * TN + frow) * 64 + (((0 * 4 + fq) ^ (frow & 7)) << 3);
    const int wa1 = (wn * 16 * TN + frow) * 64 + (((1 * 4 + fq) ^ (frow & 7)) << 3);

    f32x4 acc[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int c0 = 0, wslot = 0, hbuf = 0;
    if (wn == 0) issue_w(0, item, 0, 0);
    else issue_halo(item, 0, 0);
    while (true) {
        int nitem = item, nc0 = c0 + U;
        if (nc0 >= KC) { nc0 = 0; nitem = item + grid; }
        const bool has_next = nitem < n_items;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            // weights of this step were issued a step ago; a halo tile 4+ steps ago (the bias table, the first time, by plain stores)
#if RFD_HALO_EXP == 5 // timing only: the weight waves never wait for their DMAs (is the step waiting for the weights to land?)
            if (wn != 0 && (s == 0 || (U == 2 && s == 4))) wait_vmcnt<0>();
#else
            if (wn == 0 || s == 0 || (U == 2 && s == 4)) wait_vmcnt<0>();
#endif
#if RFD_HALO_EXP != 4
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#if RFD_HALO_EXP == 0 || RFD_HALO_EXP == 2 || RFD_HALO_EXP == 5
            if (wn == 0) {
                if (s < 8) issue_w(wslot ^ 1, item, c0, s + 1);
                else if (has_next) issue_w(wslot ^ 1, nitem, nc0, 0);
            }
#endif
#if RFD_HALO_EXP == 0 || RFD_HALO_EXP == 1 || RFD_HALO_EXP == 5
            if (wn == 1 && U == 1) {
                if (s == 0 && has_next) issue_halo(nitem, nc0, hbuf ^ 1);
            } else if (wn == 1) {
                if (s == 0) issue_halo(item, c0 + 1, 1);
                if (s == 5 && has_next) issue_halo(nitem, nc0, 0);
            }
#endif
            // ---- the step's 64 (TN = 6: 48) MFMAs as four groups, software-pipelined: the fragments of group g + 1 are read
            //      from LDS while group g's MFMAs issue (the compiler's own order was read - wait - 4 MFMAs - read - wait ...).
            //      group -> (unit k, 32-wide K half kk, first A fragment ih): TN = 4: (g >> 1, g & 1, 0); else (0, g >> 1, (g & 1) NA);
            //      a group reads NA A fragments and, when it starts a new (k, kk), the four B fragments ----
            const bf16_t *wsl = Ws + wslot * WEL;
            constexpr int NA = TN == 6 ? 3 : 4;
            bf16x8 af[2][NA], bfr[2][4];
            auto load_group = [&](int g) {
                const int k = TN == 4 ? g >> 1 : 0, kk = TN == 4 ? g & 1 : g >> 1, ih = TN == 4 ? 0 : (g & 1) * NA;
                const int u = s * U + k, cs = u / 9, tap = u - cs * 9, ky = tap / 3, kx = tap - ky * 3;
                const bf16_t *xb = Xh + (U == 2 ? cs : hbuf) * HEL + ky * HW2 * 64;
                const bf16_t *wb = wsl + k * (NWG * 64);
                if (TN == 4 || (g & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bfr[TN == 4 ? g & 1 : (g >> 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(
                            TC == 16 ? xb + j * (HW2 * 64) + (xa[0][kx] ^ (kk << 5))  // 16-wide tile: fragment j is tile row 4 wm + j
                                     : xb + (xa[j][kx] ^ (kk << 5)));
                }
#pragma unroll
                for (int i = 0; i < NA; ++i) af[g & 1][i] = *reinterpret_cast<const bf16x8 *>(wb + (ih + i) * 1024 + (kk ? wa1 : wa0));
            };
            load_group(0);
            __builtin_amdgcn_sched_group_barrier(0x100, NA + 4, 0); // group 0's reads, then the pipelined pattern below
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < 3) load_group(g + 1);
                const int ih = TN == 4 ? 0 : (g & 1) * NA, bs = TN == 4 ? g & 1 : (g >> 1) & 1;
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][i], bfr[bs][j], acc[ih + i][j], 0, 0, 0);
                // pin the interleave: one LDS read per two MFMAs while there are reads of the next group, then the rest
                if (g < 3) {
                    constexpr int kMfma = 4 * NA;
                    const int nrd = NA + ((TN == 4 || ((g + 1) & 1) == 0) ? 4 : 0);
#pragma unroll
                    for (int r = 0; r < nrd; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // DS read
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
                    }
                    if (nrd == NA) __builtin_amdgcn_sched_group_barrier(0x008, kMfma - NA, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x008, kMfma - NA - 4, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NA, 0);
                }
            }
            wslot ^= 1;
        }
        if (nc0 == 0) {
            // ---- epilogue of the item: drain first (next step's weights / next chunk's halo were issued >= a step ago), then
            //      + bias, ReLU, 16-byte stores (8 consecutive channels of one pixel per lane) ----
            wait_vmcnt<0>();
            const int t = item / tiles_n, n0 = item_n0(item), b = t / tiles_img, rem = t - b * tiles_img;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
            int frow_ = frow; // opaque: pixel coordinates are recomputed here, not kept (or spilled) across the MFMA loop
            asm volatile("" : "+v"(frow_));
#pragma unroll
            for (int ip = 0; ip < TN / 2; ++ip) {
                const int ch0 = n0 + wn * 16 * TN + ip * 32 + fq * 8;
                float bias[8];
                lds_table_read8(Tab + ch0, bias);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = wm * 64 + j * 16 + frow_, r = q / TC, c = q - r * TC;
                    const int oy = ty * TR + r, ox = tx * TC + c;
                    float o[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        o[k] = acc[2 * ip][j][k] + bias[k];
                        o[4 + k] = acc[2 * ip + 1][j][k] + bias[4 + k];
                    }
                    if (p.relu) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = fmaxf(o[k], 0.f);
                    }
                    const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                    const int nd = ch0 + p.y_coff + (ch0 >= p.y_split ? p.y_split_add : 0); // two destinations, one GEMM (SSH)
                    const uint32_t yoff = (uint32_t)(((((size_t)b * p.H + oy) * p.W + ox) * p.ldy + nd) * 2);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry,
                                                           (q < NPIX && oy < p.H && ox < p.W) ? yoff : kOob, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[2 * ip][j][k] = 0.f;
                        acc[2 * ip + 1][j][k] = 0.f;
                    }
                }
            }
        }
        if (!has_next) break;
        item = nitem;
        c0 = nc0;
        if (U == 1) hbuf ^= 1;
    }
    wait_vmcnt<0>();
}

template <int TC, int TR, int TN> static int launch_conv3x3_halo(const ConvParams &p, hipStream_t s)
{
    const int tiles_x = ceil_div(p.W, TC), tiles_y = ceil_div(p.H, TR);
    const int n_items = p.B * tiles_x * tiles_y * (p.Cout / (32 * TN));
    const int ncu = persistent_cus(p.co_running);
    const int per = ceil_div(n_items, ncu);
    const int grid = ceil_div(n_items, per);
    constexpr int HP = ((TR + 2) * (TC + 2) + 7) / 8, U = TN == 4 ? 2 : 1;
    constexpr size_t lds_need = (size_t)(2 * HP * 512 + 2 * U * 32 * TN * 64) * 2 + 2048; // two halo buffers, two weight slots, tables
    note_launch("conv3x3_halo_kernel<%d, %d, %d>", TC, TR, TN);
    return launch_persistent<conv3x3_halo_kernel<TC, TR, TN>>(grid, lds_need, s, p, tiles_x, tiles_y, n_items);
}

template <int BN, int WAVES_M, int WAVES_N>
static int launch_conv3x3_kx(const ConvParams &p, hipStream_t s)
{
    const int M = p.B * p.Ho * p.Wo;
    const int grid = ceil_div(M, 128) * (p.Cout / BN);
    const size_t lds = (size_t)(2 * 160 + 2 * BN) * 64 * sizeof(bf16_t);
    auto kern = conv3x3_kx_kernel<BN, WAVES_M, WAVES_N>;
    static DynLdsOnce once;
    RFD_TRY(once.ensure(reinterpret_cast<const void *>(kern), (int)lds));
    if (note_launch("conv3x3_kx_kernel<%d, %d, %d>", BN, WAVES_M, WAVES_N)) return RFD_OK;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES_M * WAVES_N * 64), lds, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

// ------------------------------------------------------------------------------------------------
// Long-K pointwise layers (conv1 of the units: K = 512 / 1024 -> N = 128 / 256): HBM-bound activation streaming (one pass
// over 2 K bytes per pixel for N / K <= 1/4 as many output bytes).  The generic kernel keeps ONE 16-KiB activation step in
// flight per workgroup (32 KiB per CU) and ran these layers at 2.7-3.0 TB/s.  Here a PERSISTENT workgroup owns the CU and
// walks 256-pixel x 128-channel items with a 3-slot activation ring (32 KiB steps, two steps = 64 KiB in flight) and a
// 2-slot weight ring, the streams in different waves' counters so that every wait is a plain drain of DMAs issued at least
// a step earlier (rule 1):
//   waves 0-3: weight step k + 1 (16 KiB from L2), drained at the top of every step;
//   waves 4-5: activation steps k + 2 for even k, waves 6-7 for odd k -- each pair drains only every other step, so its
//              tile has had two steps (HBM latency) to land.
// The K loop runs on across items (the next item's first tiles are requested during the last steps of this one); the
// epilogue's stores are simply queued behind the DMAs and retire with the next drain.  Same K order and MFMA sequence as
// the generic kernel: bit-identical results.
// ------------------------------------------------------------------------------------------------
#ifndef RFD_PWG_EXP
#define RFD_PWG_EXP 0 // timing experiments (tools/build_variant.sh; results are garbage)
#endif
// HAS_AFF: x' = relu(x * in_scale[c] + in_shift[c]) on the landed activation tile (the producer unit's BN + ReLU).
// WIDE: items of 128 pixels x 256 channels (waves 2 pixel x 4 channel) instead of 256 x 128 (4 x 2) -- for N = 256 layers one
// item then covers all output channels: the activation is read and, with HAS_AFF, transformed once instead of once per
// 128-channel item (stage-3 conv1: 38.8 us for an HBM floor of 13).  Same per-wave 64 x 64 tile, K order and arithmetic.
template <bool HAS_AFF, bool WIDE>
__global__ void __launch_bounds__(512) pw_gemm_kernel(const ConvParams p, int tiles_m, int n_items)
{
    RFD_CLOCK(7);
    constexpr int BM = WIDE ? 128 : 256, NWG = WIDE ? 256 : 128, XEL = BM * 64, WEL = NWG * 64;
    constexpr int WMW = BM / 64; // waves along the pixels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem); // [3][XEL]
    bf16_t *Ws = Xs + 3 * XEL;                      // [2][WEL]
    float *Tab = reinterpret_cast<float *>(Ws + 2 * WEL); // bias[Cout <= 1024] | in_scale[K <= 2048] | in_shift[K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WMW, wn = wave / WMW;
    const bool is_w = wave < 4;       // waves 0-3: weight stream, 4-7: activation stream
    const int wv = wave & 3;          // index among the weight-stream waves
    const int xgrp = (wave >> 1) & 1, xsub = wave & 1; // activation-stream waves 4-7: pair (parity of the steps it serves), half
    const int lr = lane >> 3, slot = lane & 7, frow = lane & 15, fq = lane >> 4;
    const int M = p.B * p.H * p.W, K = p.Cin, KC = K >> 6;
    const int tiles_n = p.Cout / NWG;
    const int grid = gridDim.x;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)M * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w), 0, (uint32_t)((size_t)p.Cout * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (uint32_t)((size_t)M * p.ldy * 2), 0x00020000);
    int item = xcd_remap(blockIdx.x, grid);
    if (item >= n_items) return;
    for (int c = tid; c < p.Cout; c += 512) Tab[c] = p.bias[c];
    float *Sc = Tab + p.Cout;
    if (HAS_AFF)
        for (int c = tid; c < K; c += 512) {
            Sc[c] = p.in_scale[c];
            Sc[K + c] = p.in_shift[c];
        }

    // item -> (n tile fastest, m tile): the N / 128 items of one pixel tile run next to each other on one XCD and share the
    // activation tile in its L2
    // ---- activation stream: piece (xsub + 2 q) of a step = pixels m0 + (xsub + 2 q) * 8 + lr, 16 B per lane ----
    const uint32_t xlane = (uint32_t)((((xsub * 8 + lr) * p.ldx) + p.x_coff + ((slot ^ lr) << 3)) * 2);
    auto issue_x = [&](int xslot, int it, int k) {
        const int m0 = (it / tiles_n) * BM;
#pragma unroll
        for (int q = 0; q < BM / 16; ++q) {
            const int mp = m0 + (xsub + 2 * q) * 8; // first pixel of the piece: wave-uniform (M is a multiple of 8)
            const uint32_t sb = (uint32_t)(((size_t)(m0 + 16 * q) * p.ldx) * 2 + (k << 7));
            blds16(rx, mp < M ? xlane + sb : kOob, 0, Xs + xslot * XEL + (xsub + 2 * q) * 512);
        }
    };
    // ---- weight stream: LDS row rho holds output channel perm(rho); piece (wv + 4 q) = rows (wv + 4 q) * 8 + lr ----
    uint32_t wlane;
    {
        const int rho = wv * 8 + lr, i_ = rho >> 4, fq_ = (rho >> 2) & 3, r_ = rho & 3;
        const int chn = fq_ * 8 + (i_ & 1) * 4 + r_; // + 32 q
        wlane = (uint32_t)(((size_t)chn * K + ((slot ^ lr) << 3)) * 2);
    }
    auto issue_w = [&](int wslot, int it, int k) {
        const uint32_t base = (uint32_t)((size_t)(it % tiles_n) * NWG * K * 2 + (k << 7));
#pragma unroll
        for (int q = 0; q < NWG / 32; ++q) blds16(rw, wlane, base + (uint32_t)(q * 64 * K), Ws + wslot * WEL + (wv + 4 * q) * 512);
    };
    auto advance = [&](int &it, int &k) {
        if (++k == KC) { k = 0; it += grid; }
    };

    // per-lane fragment offsets (elements): B = pixel 64 wm + 16 j + frow (swizzle key frow & 7), A = row 64 wn + 16 i + frow
    const int xb0 = (wm * 64 + frow) * 64 + ((fq ^ (frow & 7)) << 3), xb1 = xb0 ^ 32; // second 32-wide K half: chunk bit 2
    const int wa0 = (wn * 64 + frow) * 64 + ((fq ^ (frow & 7)) << 3), wa1 = wa0 ^ 32;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: X(0) by pair 0, X(1) by pair 1, W(0)
    int k = 0, xs = 0, ws = 0;
    int xi_it = item, xi_k = 0;   // next activation step to request (runs two steps ahead)
    int wi_it = item, wi_k = 0;   // next weight step to request (one step ahead)
    if (is_w) issue_w(0, item, 0);
    advance(wi_it, wi_k);
    if (!is_w && xgrp == 0) issue_x(0, xi_it, xi_k);
    advance(xi_it, xi_k);
    if (!is_w && xgrp == 1 && xi_it < n_items) issue_x(1, xi_it, xi_k);
    advance(xi_it, xi_k);
    int xis = 2; // slot of the next activation step to request
    // optional residual (FPN laterals: the coarser level, nearest-2x upsampled, added after the ReLU): requested into registers
    // with untracked buffer loads in the item's FIRST step and used in its epilogue -- every wave drains its counter at least
    // once in between (KC >= 4), the same arrangement as pw_stream's
    const int HoWo = p.Ho * p.Wo;
    const u32x4 rres = make_srd(p.res ? p.res : p.x, p.res ? (uint32_t)((size_t)(p.res_up2 ? p.B * (p.Ho >> 1) * (p.Wo >> 1) : M) * p.Cout * 2) : 0u);
    u32x4 resv[2][4];
    while (true) {
#pragma unroll
        for (int par = 0; par < 2; ++par) { // KC is even: step parity == K-step parity, the pair roles are static
            if (is_w || xgrp == par) {
                // (the residual registers were written by untracked loads in the previous step: tie them to this drain, so that
                //  the compiler cannot move or copy them before the data is there)
                if (par == 1 && k == 1 && p.res) {
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(resv[0][0]), "+v"(resv[0][1]), "+v"(resv[0][2]), "+v"(resv[0][3])::"memory");
                    asm volatile("" : "+v"(resv[1][0]), "+v"(resv[1][1]), "+v"(resv[1][2]), "+v"(resv[1][3])::"memory");
                } else {
#if RFD_PWG_EXP == 1 // timing only: the weight waves never wait (is the step waiting for the weight tile to land?)
                    if (!is_w) wait_vmcnt<0>();
#elif RFD_PWG_EXP == 2 // timing only: nobody waits
#else
                    wait_vmcnt<0>();
#endif
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (is_w) {
                if (wi_it < n_items) issue_w(ws ^ 1, wi_it, wi_k);
            } else if (xgrp == par) {
                if (xi_it < n_items) issue_x(xis, xi_it, xi_k);
            }
            if (par == 0 && k == 0 && p.res) {
                const int m0r = (item / tiles_n) * BM, n0r = (item % tiles_n) * NWG;
                int frow_ = frow;
                asm volatile("" : "+v"(frow_));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0r + wm * 64 + j * 16 + frow_;
                    size_t mr = (size_t)(m < M ? m : 0);
                    if (p.res_up2 && m < M) {
                        const int b = m / HoWo, rem = m - b * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                        mr = ((size_t)b * (p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1);
                    }
#pragma unroll
                    for (int ip = 0; ip < 2; ++ip)
                        asm_buffer_load_b128(resv[ip][j], (uint32_t)((mr * p.Cout + n0r + wn * 64 + ip * 32 + fq * 8) * 2), rres);
                }
            }
            advance(wi_it, wi_k);
            advance(xi_it, xi_k);
            xis = xis == 2 ? 0 : xis + 1;
            if (HAS_AFF) {
                // x' = relu(x * scale[c] + shift[c]) applied ONCE per element, in place in the landed tile (the same f32 multiply,
                // add, max and rounding as the generic kernel does on its fragments -- there each element is transformed by
                // both channel waves that read it, and the VALU work, not HBM, set this kernel's pace).  Thread -> logical
                // 8-channel chunk tid & 7 of rows (tid >> 3) + 64 i: one table read per step, 1 KiB contiguous per wave access.
                const int cch = tid & 7, r0 = tid >> 3;
                float ss[8], tt[8];
                lds_table_read8x2(Sc + k * 64 + cch * 8, Sc + K + k * 64 + cch * 8, ss, tt);
                bf16_t *xt = Xs + xs * XEL + r0 * 64 + ((cch ^ (r0 & 7)) << 3);
#pragma unroll
                for (int i = 0; i < BM / 64; ++i) {
                    bf16x8 v = *reinterpret_cast<const bf16x8 *>(xt + i * 64 * 64);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (__bf16)fmaxf(__builtin_fmaf((float)v[e], ss[e], tt[e]), 0.f); // one fused multiply-add: the same in every kernel that applies the input affine
                    *reinterpret_cast<bf16x8 *>(xt + i * 64 * 64) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            const bf16_t *xb = Xs + xs * XEL, *wb = Ws + ws * WEL;
            bf16x8 af[2][4], bfr[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[kk][j] = *reinterpret_cast<const bf16x8 *>(xb + j * 1024 + (kk ? xb1 : xb0));
#pragma unroll
                for (int i = 0; i < 4; ++i) af[kk][i] = *reinterpret_cast<const bf16x8 *>(wb + i * 1024 + (kk ? wa1 : wa0));
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], bfr[kk][j], acc[i][j], 0, 0, 0);
            // reads of the first half, then the second half's reads under the first half's MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
            // waves 4-5 do not drain at the top of this (odd) step: theirs comes here, inside the same loop iteration as the loads
            // (everything they have in flight -- the activation step requested a step ago -- is due at the next step's top anyway)
            if (par == 1 && k == 1 && p.res && !is_w && xgrp == 0) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(resv[0][0]), "+v"(resv[0][1]), "+v"(resv[0][2]), "+v"(resv[0][3])::"memory");
                asm volatile("" : "+v"(resv[1][0]), "+v"(resv[1][1]), "+v"(resv[1][2]), "+v"(resv[1][3])::"memory");
            }
            xs = xs == 2 ? 0 : xs + 1;
            ws ^= 1;
            ++k;
        }
        if (k == KC) {
            // ---- epilogue: + bias, ReLU, 16-byte stores; no wait -- the stores retire with the next drains ----
            const int m0 = (item / tiles_n) * BM, n0 = (item % tiles_n) * NWG;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int ch0 = n0 + wn * 64 + ip * 32 + fq * 8;
                float bias[8];
                lds_table_read8(Tab + ch0, bias);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wm * 64 + j * 16 + frow;
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = acc[2 * ip][j][e] + bias[e];
                        o[4 + e] = acc[2 * ip + 1][j][e] + bias[4 + e];
                    }
                    float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (p.res) {
                        const u32x4 rv = resv[ip][j];
                        r[0] = bf16_bits_to_f32(rv[0] & 0xffffu); r[1] = bf16_bits_to_f32(rv[0] >> 16);
                        r[2] = bf16_bits_to_f32(rv[1] & 0xffffu); r[3] = bf16_bits_to_f32(rv[1] >> 16);
                        r[4] = bf16_bits_to_f32(rv[2] & 0xffffu); r[5] = bf16_bits_to_f32(rv[2] >> 16);
                        r[6] = bf16_bits_to_f32(rv[3] & 0xffffu); r[7] = bf16_bits_to_f32(rv[3] >> 16);
                        if (!p.res_post) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] += r[e];
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
                    }
                    if (p.res && p.res_post) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += r[e];
                    }
                    const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                    const uint32_t yoff = (uint32_t)(((size_t)m * p.ldy + p.y_coff + ch0) * 2);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry, m < M ? yoff : kOob, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * ip][j][e] = 0.f;
                        acc[2 * ip + 1][j][e] = 0.f;
                    }
                }
            }
            k = 0;
            item += grid;
            if (item >= n_items) break;
        }
    }
    wait_vmcnt<0>();
}

template <bool AFF, bool WIDE> static int launch_pw_gemm_t(const ConvParams &p, hipStream_t s, int tiles_m, int n_items, int grid)
{
    // 3 activation + 2 weight slots + tables: the whole CU
    note_launch("pw_gemm_kernel<%s, %s>", AFF ? "true" : "false", WIDE ? "true" : "false");
    return launch_persistent<pw_gemm_kernel<AFF, WIDE>>(grid, kPersistentLds, s, p, tiles_m, n_items);
}

static int launch_pw_gemm(const ConvParams &p, hipStream_t s)
{
    const int M = p.B * p.H * p.W;
    // 128-pixel x 256-channel items where they cover all channels of an N = 256 layer and still fill the GPU (force_tile 15: never)
    const bool wide = p.Cout == 256 && ceil_div(M, 128) >= 150 && p.force_tile != 15;
    const int tiles_m = ceil_div(M, wide ? 128 : 256), n_items = tiles_m * (p.Cout / (wide ? 256 : 128));
    const int ncu = persistent_cus(p.co_running, true);
    const int per = ceil_div(n_items, ncu);
    const int grid = ceil_div(n_items, per);
    if (p.in_scale) return wide ? launch_pw_gemm_t<true, true>(p, s, tiles_m, n_items, grid) : launch_pw_gemm_t<true, false>(p, s, tiles_m, n_items, grid);
    return wide ? launch_pw_gemm_t<false, true>(p, s, tiles_m, n_items, grid) : launch_pw_gemm_t<false, false>(p, s, tiles_m, n_items, grid);
}

// ------------------------------------------------------------------------------------------------
// Wide pointwise GEMMs (N >= 512, K >= 384: conv3 + fused stride-2 shortcut of the first unit of stages 2-4, conv3 of
// the stage-4 units, conv1 of stage 4's first unit): MFMA-heavy, and the generic 128 x 128 tile moves 32 KiB through the
// load path and issues 8 DMA pieces per 32 MFMAs per wave.  A PERSISTENT workgroup (8 waves = 4 pixel x 2 channel, the CU
// to itself) computes 256-pixel x 256-channel items: 64 KiB and 8 pieces per 64 MFMAs per wave.  2-slot rings for both
// operands (2 x 32 + 2 x 32 KiB); every wave requests a quarter of the NEXT step's activation and weight tiles right
// after the step barrier and drains its counter at the top of the next step -- everything it waits for was issued a whole
// step (>= 64 MFMAs per wave) earlier (rule 1).  The K loop runs on across items.  Optional second K
// segment (the 1x1 stride-2 shortcut conv over x2: a per-lane pixel gather), residual, raw + activated outputs -- the
// generic kernel's epilogue arithmetic in the same order, and the same K order: bit-identical results.
// ------------------------------------------------------------------------------------------------
#ifndef RFD_WIDE_EXP
#define RFD_WIDE_EXP 0 // timing experiments (tools/build_variant.sh; results are garbage): 1 no operand DMA, 2 no DMA + no step barrier, 3 DMA issued but never waited for
#endif
__global__ void __launch_bounds__(512) pw_wide_kernel(const ConvParams p, int n_items)
{
    RFD_CLOCK(8);
    constexpr int BM = 256, NWG = 256, XEL = BM * 64, WEL = NWG * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem);        // [2][XEL]
    bf16_t *Ws = Xs + 2 * XEL;                             // [2][WEL]
    float *Tab = reinterpret_cast<float *>(Ws + 2 * WEL); // bias (+ bias2) [Cout <= 2048] | scale2 | shift2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int lr = lane >> 3, slot = lane & 7, frow = lane & 15, fq = lane >> 4;
    const int HoWo = p.Ho * p.Wo, M = p.B * HoWo;
    const int K = p.Cin + p.Cin2, KC1 = p.Cin >> 6, KC = K >> 6;
    const int tiles_n = p.Cout / NWG;
    const int grid = gridDim.x;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)M * p.ldx * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t *>(p.Cin2 ? p.x2 : p.x), 0, (uint32_t)(p.Cin2 ? (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2 : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w), 0, (uint32_t)((size_t)p.Cout * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y ? p.y : p.y2, 0, (uint32_t)((size_t)M * (p.y ? p.ldy : p.Cout) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry2 = __builtin_amdgcn_make_buffer_rsrc(p.y2 ? p.y2 : p.y, 0, (uint32_t)((size_t)M * (p.y2 ? p.Cout : p.ldy) * 2), 0x00020000);
    int item = xcd_remap(blockIdx.x, grid);
    if (item >= n_items) return;
    for (int c = tid; c < p.Cout; c += 512) {
        Tab[c] = p.bias2 ? p.bias[c] + p.bias2[c] : p.bias[c];
        if (p.y2) {
            Tab[p.Cout + c] = p.scale2[c];
            Tab[2 * p.Cout + c] = p.shift2[c];
        }
    }

    // ---- operand streams: wave w requests pieces w + 8 q (q < 4) of each 32-piece tile ----
    // activation, first segment: pixel m0 + (w + 8 q) * 8 + lr, channels 64 k ..; second: the same pixel of the stride-2 source
    const uint32_t xlane = (uint32_t)((((wave * 8 + lr) * p.ldx) + p.x_coff + ((slot ^ lr) << 3)) * 2);
    uint32_t x2off[4] = {kOob, kOob, kOob, kOob};
    auto item_x2 = [&](int it) { // per item: the lane's four source pixels of the second segment
        const int m0 = (it / tiles_n) * BM;
        int lr_ = lr;
        asm volatile("" : "+v"(lr_));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + (wave + 8 * q) * 8 + lr_;
            x2off[q] = kOob;
            if (m < M) {
                const int b = m / HoWo, rem = m - b * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
                x2off[q] = (uint32_t)(((((size_t)b * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.Cin2 + ((slot ^ lr_) << 3)) * 2);
            }
        }
    };
    uint32_t wlane;
    {
        const int rho = wave * 8 + lr, fq_ = (rho >> 2) & 3, r_ = rho & 3;
        const int chn = (rho >> 5) * 32 + fq_ * 8 + ((rho >> 4) & 1) * 4 + r_; // + 64 q
        wlane = (uint32_t)(((size_t)chn * K + ((slot ^ lr) << 3)) * 2);
    }
    auto issue = [&](int sl, int it, int k) {
        const int m0 = (it / tiles_n) * BM;
        if (k < KC1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mp = m0 + (wave + 8 * q) * 8;
                const uint32_t sb = (uint32_t)(((size_t)(m0 + 64 * q) * p.ldx) * 2 + (k << 7));
                blds16(rx, mp < M ? xlane + sb : kOob, 0, Xs + sl * XEL + (wave + 8 * q) * 512);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) blds16(rx2, x2off[q], (uint32_t)((k - KC1) << 7), Xs + sl * XEL + (wave + 8 * q) * 512);
        }
        const uint32_t wbase = (uint32_t)((size_t)(it % tiles_n) * NWG * K * 2 + (k << 7));
#pragma unroll
        for (int q = 0; q < 4; ++q) blds16(rw, wlane, wbase + (uint32_t)(q * 128 * K), Ws + sl * WEL + (wave + 8 * q) * 512);
    };

    const int xb0 = (wm * 64 + frow) * 64 + ((fq ^ (frow & 7)) << 3), xb1 = xb0 ^ 32;
    const int wa0 = (wn * 128 + frow) * 64 + ((fq ^ (frow & 7)) << 3), wa1 = wa0 ^ 32;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int k = 0, sl = 0;
    const int item0 = item;
    if (p.Cin2) item_x2(item);
    issue(0, item, 0);
    while (true) {
        int nit = item, nk = k + 1;
        if (nk == KC) { nk = 0; nit = item + grid; }
#if RFD_WIDE_EXP != 3
        wait_vmcnt<0>();
#endif
#if RFD_WIDE_EXP != 2
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        // the next item's second-segment pixels: needed from its step KC1 on, so recomputed one step into the item
        if (p.Cin2 && k == 0 && item != item0) item_x2(item);
#if RFD_WIDE_EXP == 0 || RFD_WIDE_EXP == 3
        if (nit < n_items) issue(sl ^ 1, nit, nk);
#endif
        const bf16_t *xb = Xs + sl * XEL, *wb = Ws + sl * WEL;
        bf16x8 af[2][4], bfr[2][4];
        auto load_group = [&](int g) { // group g -> (kk = g >> 1, A fragments 4 (g & 1) ..); B fragments with the first of a kk
            const int kk = g >> 1, ih = (g & 1) * 4;
            if ((g & 1) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[kk][j] = *reinterpret_cast<const bf16x8 *>(xb + j * 1024 + (kk ? xb1 : xb0));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) af[g & 1][i] = *reinterpret_cast<const bf16x8 *>(wb + (ih + i) * 1024 + (kk ? wa1 : wa0));
        };
        load_group(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) load_group(g + 1);
            const int ih = (g & 1) * 4, kk = g >> 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][i], bfr[kk][j], acc[ih + i][j], 0, 0, 0);
            if (g < 3) {
                const int nrd = 4 + (((g + 1) & 1) == 0 ? 4 : 0);
#pragma unroll
                for (int r = 0; r < nrd; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                if (nrd == 4) __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                else __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        }
        sl ^= 1;
        if (nk == 0) {
            // ---- epilogue of the item (the generic kernel's arithmetic, in its order) ----
            const int m0 = (item / tiles_n) * BM, n0 = (item % tiles_n) * NWG;
            int frow_ = frow;
            asm volatile("" : "+v"(frow_));
#pragma unroll
            for (int ip = 0; ip < 4; ++ip) {
                const int ch0 = n0 + wn * 128 + ip * 32 + fq * 8;
                uint4 rv[4];
                if (p.res) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = m0 + wm * 64 + j * 16 + frow_;
                        rv[j] = *reinterpret_cast<const uint4 *>(p.res + (size_t)(m < M ? m : 0) * p.Cout + ch0);
                    }
                }
                float bias[8], s2[8], t2[8];
                if (p.y2) lds_table_read8x3(Tab + ch0, Tab + p.Cout + ch0, Tab + 2 * p.Cout + ch0, bias, s2, t2);
                else lds_table_read8(Tab + ch0, bias);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = m0 + wm * 64 + j * 16 + frow_;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[2 * ip][j][e] + bias[e];
                        v[4 + e] = acc[2 * ip + 1][j][e] + bias[4 + e];
                        acc[2 * ip][j][e] = 0.f;
                        acc[2 * ip + 1][j][e] = 0.f;
                    }
                    if (p.res) {
                        const uint4 r4 = rv[j];
                        const float r[8] = {bf16_bits_to_f32(r4.x & 0xffffu), bf16_bits_to_f32(r4.x >> 16), bf16_bits_to_f32(r4.y & 0xffffu),
                                            bf16_bits_to_f32(r4.y >> 16), bf16_bits_to_f32(r4.z & 0xffffu), bf16_bits_to_f32(r4.z >> 16),
                                            bf16_bits_to_f32(r4.w & 0xffffu), bf16_bits_to_f32(r4.w >> 16)};
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
                    if (p.y) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = p.relu ? fmaxf(v[e], 0.f) : v[e];
                        const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                        const uint32_t yoff = (uint32_t)(((size_t)m * p.ldy + p.y_coff + ch0) * 2);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry, m < M ? yoff : kOob, 0, 0);
                    }
                    if (p.y2) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = fmaxf(v[e] * s2[e] + t2[e], 0.f);
                        const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                        const uint32_t yoff = (uint32_t)(((size_t)m * p.Cout + ch0) * 2);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{lo.x, lo.y, hi.x, hi.y}, ry2, m < M ? yoff : kOob, 0, 0);
                    }
                }
            }
        }
        if (nit >= n_items) break;
        item = nit;
        k = nk;
    }
    wait_vmcnt<0>();
}

static int launch_pw_wide(const ConvParams &p, hipStream_t s)
{
    const int M = p.B * p.Ho * p.Wo;
    const int n_items = ceil_div(M, 256) * (p.Cout / 256);
    const int ncu = persistent_cus(p.co_running);
    const int per = ceil_div(n_items, ncu);
    const int grid = ceil_div(n_items, per);
    // 4 x 32 KiB + tables: the whole CU
    note_launch("pw_wide_kernel");
    return launch_persistent<pw_wide_kernel>(grid, kPersistentLds, s, p, n_items);
}

// ------------------------------------------------------------------------------------------------
// Back-to-back fusion for the pre-activation units of stage 1:
//     raw = conv3(t2) [+ shortcut(act)] + bias (+ residual)      -> HBM (the next unit's residual)
//     a   = relu(raw * scale + shift)                             -> bf16 operand tile in LDS only
//     t1' = relu(conv1_next(a) + bias1)                           -> HBM
// One 8-wave workgroup owns 128 pixels x all 256 channels, so the 1x1 "next conv1" (K = 256) can run on
// the activated tile without it ever leaving the CU: the 419 MB tensor is written once (raw) and no longer
// read by conv1.  Everything is resident in LDS: X k-tiles (16 KiB each, up to 2), W3 k-tiles (32 KiB each),
// the 128 x 256 bf16 operand tile (64 KiB), W1 (32 KiB) -- 144 KiB for K1 = 64; for K1 = 128 (unit 1: fused
// shortcut) the second X/W3 k-tile reuses the first one's slot behind a barrier.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) conv_b2b_s1_kernel(const B2BParams p)
{
    RFD_CLOCK(9);
    constexpr int BM = 128, N1 = 256, N2 = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem);  // [128][64]
    bf16_t *W3s = Xs + BM * 64;                      // [256][64]
    bf16_t *A2 = W3s + N1 * 64;                      // [4][128][64]  k-tiles of the activated tile
    bf16_t *W1s = A2 + 4 * BM * 64;                  // [4][64][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;        // phase 1: 2 (m) x 4 (n) waves, 64 x 64 each
    const int HW = p.H * p.W, M = p.B * HW;
    const int K1 = p.Cin + p.Cin2, nk1 = K1 >> 6;
    const int m0 = blockIdx.x * BM;
    const int lr = lane >> 3, chunk = (lane & 7) ^ lr, frow = lane & 15, fq = lane >> 4;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)M * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.Cin2 ? p.x2 : p.x), 0,
                                                                         (uint32_t)(p.Cin2 ? (size_t)M * p.Cin2 * 2 : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w3), 0, (uint32_t)((size_t)N1 * K1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w1), 0, (uint32_t)((size_t)N2 * N1 * 2), 0x00020000);

    // permuted channel map of an operand-A row (see conv_igemm_kernel): row rho of a 64-wide slice
    auto perm64 = [](int rho) { const int i_ = rho >> 4, fq_ = (rho >> 2) & 3, r_ = rho & 3; return (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_; };
    // ---- stage: X tile kt (2 pieces / wave), W3 tile kt (4 pieces / wave), and once W1 (4 pieces / wave) ----
    auto stage1 = [&](int kt) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = wave + 8 * q, m = m0 + piece * 8 + lr;
            uint32_t off = kOob;
            if (m < M) off = (uint32_t)(((size_t)m * (kt < (p.Cin >> 6) ? p.Cin : p.Cin2) + chunk * 8) * 2);
            if (kt < (p.Cin >> 6)) blds16(rx, off, (uint32_t)(kt << 7), Xs + piece * 512);
            else blds16(rx2, off, (uint32_t)((kt - (p.Cin >> 6)) << 7), Xs + piece * 512);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave + 8 * q, rho = piece * 8 + lr; // 0..255
            const int chn = (rho & ~63) + perm64(rho & 63);
            blds16(rw3, (uint32_t)(((size_t)chn * K1 + chunk * 8) * 2), (uint32_t)(kt << 7), W3s + piece * 512);
        }
    };
    stage1(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { // W1: [4 k-tiles][64 rows][64]: piece = kt2*8 + (row/8)
        const int piece = wave + 8 * q, kt2 = piece >> 3, rho = (piece & 7) * 8 + lr;
        blds16(rw1, (uint32_t)(((size_t)perm64(rho) * N1 + chunk * 8) * 2), (uint32_t)(kt2 << 7), W1s + piece * 512);
    }
    // residual prefetch for this wave's 64 x 64 output slice: [j][h] = pixel j*16+frow, channels h*32+fq*8..+7
    uint4 resv[4][2];
    if (p.res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + frow;
            const size_t mr = (size_t)(m < M ? m : 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) resv[j][h] = *reinterpret_cast<const uint4 *>(p.res + mr * N1 + wn * 64 + h * 32 + fq * 8);
        }
    }

    // ---- phase 1: acc1[64 n x 64 m per wave] over K1 ----
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk1; ++kt) {
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* no LDS read in flight at a barrier that frees a ring slot for DMA (tools/isa_check.py) */
        const bf16_t *xs = Xs + (wm * 64) * 64, *ws = W3s + (wn * 64) * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
            const int ch = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int r = i * 16 + frow; af[i] = *reinterpret_cast<const bf16x8 *>(ws + r * 64 + ((ch ^ (r & 7)) << 3)); }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int r = j * 16 + frow; bfr[j] = *reinterpret_cast<const bf16x8 *>(xs + r * 64 + ((ch ^ (r & 7)) << 3)); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk1) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* no LDS read in flight at a barrier that frees a ring slot for DMA (tools/isa_check.py) */ // every wave is done with the slot
            stage1(kt + 1);
        }
    }

    // ---- epilogue 1: raw -> HBM, relu(affine(raw)) -> LDS operand tile (k-tile = wn, chunk = h*4 + fq) ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = wn * 64 + h * 32 + fq * 8;
        float bias[8], s2[8], t2[8];
        {
            const float4 b0 = *reinterpret_cast<const float4 *>(p.bias3 + n), b1 = *reinterpret_cast<const float4 *>(p.bias3 + n + 4);
            bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
            if (p.bias3b) {
                const float4 d0 = *reinterpret_cast<const float4 *>(p.bias3b + n), d1 = *reinterpret_cast<const float4 *>(p.bias3b + n + 4);
                bias[0] += d0.x; bias[1] += d0.y; bias[2] += d0.z; bias[3] += d0.w; bias[4] += d1.x; bias[5] += d1.y; bias[6] += d1.z; bias[7] += d1.w;
            }
            const float4 a0 = *reinterpret_cast<const float4 *>(p.scale + n), a1 = *reinterpret_cast<const float4 *>(p.scale + n + 4);
            const float4 c0 = *reinterpret_cast<const float4 *>(p.shift + n), c1 = *reinterpret_cast<const float4 *>(p.shift + n + 4);
            s2[0] = a0.x; s2[1] = a0.y; s2[2] = a0.z; s2[3] = a0.w; s2[4] = a1.x; s2[5] = a1.y; s2[6] = a1.z; s2[7] = a1.w;
            t2[0] = c0.x; t2[1] = c0.y; t2[2] = c0.z; t2[3] = c0.w; t2[4] = c1.x; t2[5] = c1.y; t2[6] = c1.z; t2[7] = c1.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wm * 64 + j * 16 + frow, m = m0 + row;
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = acc[2 * h][j][k] + bias[k]; v[4 + k] = acc[2 * h + 1][j][k] + bias[4 + k]; }
            if (p.res) {
                const uint4 rv = resv[j][h];
                v[0] += bf16_bits_to_f32(rv.x & 0xffffu); v[1] += bf16_bits_to_f32(rv.x >> 16);
                v[2] += bf16_bits_to_f32(rv.y & 0xffffu); v[3] += bf16_bits_to_f32(rv.y >> 16);
                v[4] += bf16_bits_to_f32(rv.z & 0xffffu); v[5] += bf16_bits_to_f32(rv.z >> 16);
                v[6] += bf16_bits_to_f32(rv.w & 0xffffu); v[7] += bf16_bits_to_f32(rv.w >> 16);
            }
            const uint2 lo = pack_bf16x4(v[0], v[1], v[2], v[3]), hi = pack_bf16x4(v[4], v[5], v[6], v[7]);
            if (m < M) *reinterpret_cast<uint4 *>(p.raw + (size_t)m * N1 + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            // the consumer conv1 sees relu(affine(bf16(raw))): same rounding points as the unfused pair
            float a[8];
            const uint32_t rb[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[2 * k] = fmaxf(bf16_bits_to_f32(rb[k] & 0xffffu) * s2[2 * k] + t2[2 * k], 0.f);
                a[2 * k + 1] = fmaxf(bf16_bits_to_f32(rb[k] >> 16) * s2[2 * k + 1] + t2[2 * k + 1], 0.f);
            }
            const uint2 alo = pack_bf16x4(a[0], a[1], a[2], a[3]), ahi = pack_bf16x4(a[4], a[5], a[6], a[7]);
            const int c = h * 4 + fq;
            *reinterpret_cast<uint4 *>(A2 + wn * BM * 64 + row * 64 + ((c ^ (row & 7)) << 3)) = make_uint4(alo.x, alo.y, ahi.x, ahi.y);
        }
    }
    wait_vmcnt<0>(); // W1 has landed (issued first)
    __syncthreads();

    // ---- phase 2: t1[16 pixels x 64 channels per wave] = A2[128 x 256] . W1^T ----
    f32x4 acc2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt2 = 0; kt2 < 4; ++kt2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = kk * 4 + fq;
            const int r = wave * 16 + frow;
            const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(A2 + kt2 * BM * 64 + r * 64 + ((ch ^ (r & 7)) << 3));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = i * 16 + frow;
                const bf16x8 af = *reinterpret_cast<const bf16x8 *>(W1s + kt2 * N2 * 64 + ra * 64 + ((ch ^ (ra & 7)) << 3));
                acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc2[i], 0, 0, 0);
            }
        }
    }
    const int m = m0 + wave * 16 + frow;
    if (m < M) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = h * 32 + fq * 8;
            const float4 b0 = *reinterpret_cast<const float4 *>(p.bias1 + n), b1 = *reinterpret_cast<const float4 *>(p.bias1 + n + 4);
            const uint2 lo = pack_bf16x4(fmaxf(acc2[2 * h][0] + b0.x, 0.f), fmaxf(acc2[2 * h][1] + b0.y, 0.f), fmaxf(acc2[2 * h][2] + b0.z, 0.f), fmaxf(acc2[2 * h][3] + b0.w, 0.f));
            const uint2 hi = pack_bf16x4(fmaxf(acc2[2 * h + 1][0] + b1.x, 0.f), fmaxf(acc2[2 * h + 1][1] + b1.y, 0.f), fmaxf(acc2[2 * h + 1][2] + b1.z, 0.f), fmaxf(acc2[2 * h + 1][3] + b1.w, 0.f));
            *reinterpret_cast<uint4 *>(p.t1 + (size_t)m * N2 + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
}

// Persistent form of the back-to-back kernel for K1 = 64 (no fused shortcut: units 2.. of stage 1).  The kernel above
// re-stages W3 (32 KiB) and W1 (32 KiB) for every 128-pixel tile -- as many bytes as the tile's activation and residual --
// and runs its five phases strictly one after the other: 11 us per tile where the tile's HBM traffic (160 KiB) needs 6.
// Here a workgroup owns the CU, keeps both filter banks in LDS for its lifetime and walks tiles: the NEXT tile's activation
// tile is requested right after phase 1 has read the current one (it lands under epilogue 1 + phase 2), the next tile's
// residual after epilogue 1 has consumed the current one.  Drain-only waits (DESIGN.md section 5 rule 1): the one at the top
// of a tile covers DMAs and stores issued at least a phase earlier.  Same arithmetic in the same order: bit-identical.
__global__ void __launch_bounds__(512) conv_b2b_s1_persistent_kernel(const B2BParams p, int ntiles)
{
    RFD_CLOCK(10);
    constexpr int BM = 128, N1 = 256, N2 = 64, K1 = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem);  // [128][64]
    bf16_t *W3s = Xs + BM * 64;                      // [256][64]   resident
    bf16_t *A2 = W3s + N1 * 64;                      // [4][128][64] k-tiles of the activated tile
    bf16_t *W1s = A2 + 4 * BM * 64;                  // [4][64][64] resident
    // bias3 | scale | shift [256 each] | bias1 [64]: in LDS, so that the epilogues issue no global loads -- a load issued behind the
    // next tile's activation DMA could only be waited for together with it
    float *Tab = reinterpret_cast<float *>(W1s + 4 * N2 * 64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1; // phase 1: 2 (m) x 4 (n) waves, 64 x 64 each
    const int M = p.B * p.H * p.W;
    const int lr = lane >> 3, chunk = (lane & 7) ^ lr, frow = lane & 15, fq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)M * K1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w3), 0, (uint32_t)((size_t)N1 * K1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w1), 0, (uint32_t)((size_t)N2 * N1 * 2), 0x00020000);
    auto perm64 = [](int rho) { const int i_ = rho >> 4, fq_ = (rho >> 2) & 3, r_ = rho & 3; return (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_; };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    for (int c = tid; c < N1; c += 512) {
        Tab[c] = p.bias3[c];
        Tab[N1 + c] = p.scale[c];
        Tab[2 * N1 + c] = p.shift[c];
        if (c < N2) Tab[3 * N1 + c] = p.bias1[c];
    }

    auto stage_x = [&](int t) { // activation tile of tile t: 2 pieces / wave
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = wave + 8 * q, m = t * BM + piece * 8 + lr;
            blds16(rx, m < M ? (uint32_t)(((size_t)m * K1 + chunk * 8) * 2) : kOob, 0, Xs + piece * 512);
        }
    };
    uint4 resv[4][2]; // residual of this wave's 64 x 64 output slice: [j][h] = pixel j*16+frow, channels h*32+fq*8..+7
    auto load_res = [&](int t) {
        if (!p.res) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = t * BM + wm * 64 + j * 16 + frow;
            const size_t mr = (size_t)(m < M ? m : 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) resv[j][h] = *reinterpret_cast<const uint4 *>(p.res + mr * N1 + wn * 64 + h * 32 + fq * 8);
        }
    };
    // once: W3 (4 pieces / wave), W1 (4 pieces / wave: [4 k-tiles][64 rows][64], piece = kt2*8 + row/8)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave + 8 * q, rho = piece * 8 + lr; // 0..255
        const int chn = (rho & ~63) + perm64(rho & 63);
        blds16(rw3, (uint32_t)(((size_t)chn * K1 + chunk * 8) * 2), 0, W3s + piece * 512);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave + 8 * q, kt2 = piece >> 3, rho = (piece & 7) * 8 + lr;
        blds16(rw1, (uint32_t)(((size_t)perm64(rho) * N1 + chunk * 8) * 2), (uint32_t)(kt2 << 7), W1s + piece * 512);
    }
    stage_x(tile);
    load_res(tile);

    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * BM, next = tile + gridDim.x;
        // this tile's activation (requested a phase 2 + ... ago; the first time: + the filter banks) is in LDS in every wave
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- phase 1: acc1[64 n x 64 m per wave], one K step ----
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const bf16_t *xs = Xs + (wm * 64) * 64, *ws = W3s + (wn * 64) * 64;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[4], bfr[4];
                const int ch = kk * 4 + fq;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int r = i * 16 + frow; af[i] = *reinterpret_cast<const bf16x8 *>(ws + r * 64 + ((ch ^ (r & 7)) << 3)); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int r = j * 16 + frow; bfr[j] = *reinterpret_cast<const bf16x8 *>(xs + r * 64 + ((ch ^ (r & 7)) << 3)); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        // every wave has read Xs (and, from the previous tile's phase 2, A2): the next activation tile may land
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (next < ntiles) stage_x(next);

        // ---- epilogue 1: raw -> HBM, relu(affine(raw)) -> LDS operand tile (k-tile = wn, chunk = h*4 + fq) ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = wn * 64 + h * 32 + fq * 8;
            float bias[8], s2[8], t2[8];
            lds_table_read8x3(Tab + n, Tab + N1 + n, Tab + 2 * N1 + n, bias, s2, t2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + frow, m = m0 + row;
                float v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = acc[2 * h][j][k] + bias[k]; v[4 + k] = acc[2 * h + 1][j][k] + bias[4 + k]; }
                if (p.res) {
                    const uint4 rv = resv[j][h];
                    v[0] += bf16_bits_to_f32(rv.x & 0xffffu); v[1] += bf16_bits_to_f32(rv.x >> 16);
                    v[2] += bf16_bits_to_f32(rv.y & 0xffffu); v[3] += bf16_bits_to_f32(rv.y >> 16);
                    v[4] += bf16_bits_to_f32(rv.z & 0xffffu); v[5] += bf16_bits_to_f32(rv.z >> 16);
                    v[6] += bf16_bits_to_f32(rv.w & 0xffffu); v[7] += bf16_bits_to_f32(rv.w >> 16);
                }
                const uint2 lo = pack_bf16x4(v[0], v[1], v[2], v[3]), hi = pack_bf16x4(v[4], v[5], v[6], v[7]);
                if (m < M) *reinterpret_cast<uint4 *>(p.raw + (size_t)m * N1 + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                // the consumer conv1 sees relu(affine(bf16(raw))): same rounding points as the unfused pair
                float a[8];
                const uint32_t rb[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[2 * k] = fmaxf(bf16_bits_to_f32(rb[k] & 0xffffu) * s2[2 * k] + t2[2 * k], 0.f);
                    a[2 * k + 1] = fmaxf(bf16_bits_to_f32(rb[k] >> 16) * s2[2 * k + 1] + t2[2 * k + 1], 0.f);
                }
                const uint2 alo = pack_bf16x4(a[0], a[1], a[2], a[3]), ahi = pack_bf16x4(a[4], a[5], a[6], a[7]);
                const int c = h * 4 + fq;
                *reinterpret_cast<uint4 *>(A2 + wn * BM * 64 + row * 64 + ((c ^ (row & 7)) << 3)) = make_uint4(alo.x, alo.y, ahi.x, ahi.y);
            }
        }
        if (next < ntiles) load_res(next); // consumed above; lands under phase 2 and the next tile's phase 1
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the operand tile is complete (LDS only: no DMA wait here)

        // ---- phase 2: t1[16 pixels x 64 channels per wave] = A2[128 x 256] . W1^T ----
        f32x4 acc2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt2 = 0; kt2 < 4; ++kt2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ch = kk * 4 + fq;
                const int r = wave * 16 + frow;
                const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(A2 + kt2 * BM * 64 + r * 64 + ((ch ^ (r & 7)) << 3));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = i * 16 + frow;
                    const bf16x8 af = *reinterpret_cast<const bf16x8 *>(W1s + kt2 * N2 * 64 + ra * 64 + ((ch ^ (ra & 7)) << 3));
                    acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc2[i], 0, 0, 0);
                }
            }
        }
        const int m = m0 + wave * 16 + frow;
        {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = h * 32 + fq * 8;
                float b1v[8];
                lds_table_read8(Tab + 3 * N1 + n, b1v);
                const uint2 lo = pack_bf16x4(fmaxf(acc2[2 * h][0] + b1v[0], 0.f), fmaxf(acc2[2 * h][1] + b1v[1], 0.f), fmaxf(acc2[2 * h][2] + b1v[2], 0.f), fmaxf(acc2[2 * h][3] + b1v[3], 0.f));
                const uint2 hi = pack_bf16x4(fmaxf(acc2[2 * h + 1][0] + b1v[4], 0.f), fmaxf(acc2[2 * h + 1][1] + b1v[5], 0.f), fmaxf(acc2[2 * h + 1][2] + b1v[6], 0.f), fmaxf(acc2[2 * h + 1][3] + b1v[7], 0.f));
                if (m < M) *reinterpret_cast<uint4 *>(p.t1 + (size_t)m * N2 + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
}

// Persistent form for K1 = 128 (unit 1 of stage 1: conv3 + the fused 1x1 shortcut, no residual).  That unit moves half the bytes
// of the others and still took as long in the one-tile-per-workgroup kernel (1.4 TB/s): per 128-pixel tile it re-staged 96 KiB
// of filter banks and ran its phases one after the other.  Both banks (64 + 32 KiB) only fit LDS next to the operand tile if
// the tile is 64 pixels: Xs 16 + W3 64 + A2 32 + W1 32 = 144 KiB.  Phase 1: 8 waves x (64 pixels x 32 channels); phase 2:
// 4 x 2 waves x (16 pixels x 32 channels).  Same K order and arithmetic as conv_b2b_s1_kernel: bit-identical.
__global__ void __launch_bounds__(512) conv_b2b_s1_persistent_k128_kernel(const B2BParams p, int ntiles)
{
    RFD_CLOCK(11);
    constexpr int BM = 64, N1 = 256, N2 = 64, K1 = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem);  // [2 k-tiles][64][64]
    bf16_t *W3s = Xs + 2 * BM * 64;                  // [2 k-tiles][256][64] resident
    bf16_t *A2 = W3s + 2 * N1 * 64;                  // [4 k-tiles][64][64]  the activated tile
    bf16_t *W1s = A2 + 4 * BM * 64;                  // [4 k-tiles][64][64]  resident
    float *Tab = reinterpret_cast<float *>(W1s + 4 * N2 * 64); // bias3 + bias3b | scale | shift [256 each] | bias1 [64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.B * p.H * p.W;
    const int lr = lane >> 3, chunk = (lane & 7) ^ lr, frow = lane & 15, fq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)M * 64 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.x2), 0, (uint32_t)((size_t)M * 64 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w3), 0, (uint32_t)((size_t)N1 * K1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(p.w1), 0, (uint32_t)((size_t)N2 * N1 * 2), 0x00020000);
    auto perm64 = [](int rho) { const int i_ = rho >> 4, fq_ = (rho >> 2) & 3, r_ = rho & 3; return (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_; };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    for (int c = tid; c < N1; c += 512) {
        Tab[c] = p.bias3b ? p.bias3[c] + p.bias3b[c] : p.bias3[c];
        Tab[N1 + c] = p.scale[c];
        Tab[2 * N1 + c] = p.shift[c];
        if (c < N2) Tab[3 * N1 + c] = p.bias1[c];
    }
    auto stage_x = [&](int t) { // 16 pieces: k-tile 0 from x, k-tile 1 from the shortcut's input x2; 2 per wave
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = wave + 8 * q, kt = piece >> 3, m = t * BM + (piece & 7) * 8 + lr;
            const uint32_t off = m < M ? (uint32_t)(((size_t)m * 64 + chunk * 8) * 2) : kOob;
            if (kt == 0) blds16(rx, off, 0, Xs + piece * 512);
            else blds16(rx2, off, 0, Xs + piece * 512);
        }
    };
    // once: W3 (2 k-tiles x 32 pieces: 8 / wave), W1 (32 pieces: 4 / wave)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int piece = wave + 8 * q, kt = piece >> 5, rho = (piece & 31) * 8 + lr; // row 0..255
        const int chn = (rho & ~63) + perm64(rho & 63);
        blds16(rw3, (uint32_t)(((size_t)chn * K1 + chunk * 8) * 2), (uint32_t)(kt << 7), W3s + piece * 512);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave + 8 * q, kt2 = piece >> 3, rho = (piece & 7) * 8 + lr;
        blds16(rw1, (uint32_t)(((size_t)perm64(rho) * N1 + chunk * 8) * 2), (uint32_t)(kt2 << 7), W1s + piece * 512);
    }
    stage_x(tile);

    const int sw0 = (fq ^ (frow & 7)) << 3, sw1 = sw0 ^ 32; // chunk position of the two 32-wide K halves for rows = frow mod 8
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * BM, next = tile + gridDim.x;
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- phase 1: acc[2 x 16 channels][4 x 16 pixels] per wave over K1 = 2 k-tiles x 2 halves ----
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[2], bfr[4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[i] = *reinterpret_cast<const bf16x8 *>(W3s + kt * N1 * 64 + (wave * 32 + i * 16 + frow) * 64 + (kk ? sw1 : sw0));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8 *>(Xs + kt * BM * 64 + (j * 16 + frow) * 64 + (kk ? sw1 : sw0));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        // every wave has read Xs (and, from the previous tile's phase 2, A2): the next activation tile may land
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (next < ntiles) stage_x(next);

        // ---- epilogue 1: raw -> HBM, relu(affine(bf16(raw))) -> the LDS operand tile ----
        {
            const int n = wave * 32 + fq * 8; // this lane's 8 consecutive channels
            float bias[8], s2[8], t2[8];
            lds_table_read8x3(Tab + n, Tab + N1 + n, Tab + 2 * N1 + n, bias, s2, t2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = j * 16 + frow, m = m0 + row;
                float v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = acc[0][j][k] + bias[k]; v[4 + k] = acc[1][j][k] + bias[4 + k]; }
                const uint2 lo = pack_bf16x4(v[0], v[1], v[2], v[3]), hi = pack_bf16x4(v[4], v[5], v[6], v[7]);
                if (m < M) *reinterpret_cast<uint4 *>(p.raw + (size_t)m * N1 + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                float a[8];
                const uint32_t rb[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[2 * k] = fmaxf(bf16_bits_to_f32(rb[k] & 0xffffu) * s2[2 * k] + t2[2 * k], 0.f);
                    a[2 * k + 1] = fmaxf(bf16_bits_to_f32(rb[k] >> 16) * s2[2 * k + 1] + t2[2 * k + 1], 0.f);
                }
                const uint2 alo = pack_bf16x4(a[0], a[1], a[2], a[3]), ahi = pack_bf16x4(a[4], a[5], a[6], a[7]);
                const int c = (wave & 1) * 4 + fq; // 8-channel chunk inside k-tile wave >> 1
                *reinterpret_cast<uint4 *>(A2 + (wave >> 1) * BM * 64 + row * 64 + ((c ^ (row & 7)) << 3)) = make_uint4(alo.x, alo.y, ahi.x, ahi.y);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the operand tile is complete (LDS only)

        // ---- phase 2: t1[16 pixels x 32 channels per wave] = A2[64 x 256] . W1^T ----
        f32x4 acc2[2];
        acc2[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int pr = (wave & 3) * 16 + frow, wh = wave >> 2;
#pragma unroll
        for (int kt2 = 0; kt2 < 4; ++kt2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(A2 + kt2 * BM * 64 + pr * 64 + (kk ? sw1 : sw0));
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8 *>(W1s + kt2 * N2 * 64 + (wh * 32 + i * 16 + frow) * 64 + (kk ? sw1 : sw0));
                    acc2[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc2[i], 0, 0, 0);
                }
            }
        }
        {
            const int m = m0 + pr, n = wh * 32 + fq * 8;
            float b1v[8];
            lds_table_read8(Tab + 3 * N1 + n, b1v);
            const uint2 lo = pack_bf16x4(fmaxf(acc2[0][0] + b1v[0], 0.f), fmaxf(acc2[0][1] + b1v[1], 0.f), fmaxf(acc2[0][2] + b1v[2], 0.f), fmaxf(acc2[0][3] + b1v[3], 0.f));
            const uint2 hi = pack_bf16x4(fmaxf(acc2[1][0] + b1v[4], 0.f), fmaxf(acc2[1][1] + b1v[5], 0.f), fmaxf(acc2[1][2] + b1v[6], 0.f), fmaxf(acc2[1][3] + b1v[7], 0.f));
            if (m < M) *reinterpret_cast<uint4 *>(p.t1 + (size_t)m * N2 + n) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
}

int launch_conv_b2b_s1(const B2BParams &p, hipStream_t s)
{
    if (p.Cin != 64 || (p.Cin2 != 0 && p.Cin2 != 64)) {
        set_error("b2b: unsupported shape Cin=%d Cin2=%d", p.Cin, p.Cin2);
        return RFD_ERR_INVALID_ARG;
    }
    const int M = p.B * p.H * p.W;
    const size_t lds = (size_t)(128 * 64 + 256 * 64 + 4 * 128 * 64 + 4 * 64 * 64) * sizeof(bf16_t); // 144 KiB
    const int ntiles = ceil_div(M, 128);
    // Round 4: the weight-resident barrier-free pair kernel (pw_pair_kernel<.., NCR = 2, HALF1>, the form that runs the stage
    // 1 -> 2 boundary) takes stage 1's pairs too: W3 [256][64 or 128] + W1 [64][256] stay in LDS, every wave is an independent
    // pipeline over its 16 pixels.  force_tile 16 forces it, 6 keeps the older persistent kernels below (RFD_S1_PAIR=0: always).
    static const int s1_pair = [] { const char *e = getenv("RFD_S1_PAIR"); return e ? atoi(e) : 1; }();
    const bool pair_ok = (const char *)p.w1 > (const char *)p.w3 && (size_t)((const char *)p.w1 - (const char *)p.w3) < (1u << 30) &&
                         (p.Cin2 == 0 ? p.res != nullptr : (!p.res && p.bias3b)) && (size_t)M * 256 * 2 < 0xfffffff0ull;
    if (pair_ok && (p.force_tile == 16 || (p.force_tile == 0 && s1_pair && ntiles >= 512))) {
        ConvParams c;
        memset(&c, 0, sizeof c);
        c.x = p.x; c.w = p.w3; c.bias = p.bias3; c.x2 = p.x2; c.bias2 = p.bias3b; c.res = p.res;
        c.scale2 = p.scale; c.shift2 = p.shift; c.y = p.raw; c.w1 = p.w1; c.bias1 = p.bias1; c.t1 = p.t1; c.n1 = 64;
        c.B = p.B; c.H = c.Ho = c.H2 = p.H; c.W = c.Wo = c.W2 = p.W; c.Cin = p.Cin; c.Cin2 = p.Cin2; c.stride2 = 1; c.Cout = 256;
        c.KH = c.KW = 1; c.stride = 1; c.ldx = p.Cin; c.ldy = 256; c.y_split = c.n_valid = 1 << 30; c.co_running = 1;
        return p.Cin2 ? launch_pw_pair<1, 1, false, 1, 2, true>(c, s) : launch_pw_pair<1, 1, false, 0, 2, true>(c, s);
    }
    // K1 = 64 and at least two tiles per CU: the persistent form (force_tile 7 opts out, 6 forces it whatever the size)
    if (p.Cin2 == 0 && p.force_tile != 7 && p.force_tile != 1 && p.force_tile != 2 && (ntiles >= 512 || p.force_tile == 6)) {
        const int per = ceil_div(ntiles, persistent_cus(1, true));
        const int grid = ceil_div(ntiles, per);
        note_launch("conv_b2b_s1_persistent_kernel");
        return launch_persistent<conv_b2b_s1_persistent_kernel>(grid, kPersistentLds, s, p, ntiles);
    }
    // K1 = 128 (fused shortcut, no residual): persistent 64-pixel tiles, both filter banks resident
    if (p.Cin2 == 64 && !p.res && p.force_tile != 7 && p.force_tile != 1 && p.force_tile != 2 && (M >= 64 * 1024 || p.force_tile == 6)) {
        const int nt = ceil_div(M, 64), per = ceil_div(nt, persistent_cus(1, true));
        note_launch("conv_b2b_s1_persistent_k128_kernel");
        return launch_persistent<conv_b2b_s1_persistent_k128_kernel>(ceil_div(nt, per), kPersistentLds, s, p, nt);
    }
    static DynLdsOnce once;
    RFD_TRY(once.ensure(reinterpret_cast<const void *>(conv_b2b_s1_kernel), (int)lds));
    if (note_launch("conv_b2b_s1_kernel")) return RFD_OK;
    hipLaunchKernelGGL(conv_b2b_s1_kernel, dim3(ceil_div(M, 128)), dim3(512), lds, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSX, bool CHUNK_MAJOR>
static int launch_conv_cfg_order(const ConvParams &p, hipStream_t s)
{
    const int M = p.B * p.Ho * p.Wo;
    const int grid = ceil_div(M, BM) * (p.Cout / BN);
    const int nk = (p.KH * p.KW * p.Cin + p.Cin2) / 64;
    const size_t full = (size_t)(NSX * BM + 2 * BN) * 64 * sizeof(bf16_t);
    const size_t aff_bytes = p.in_scale ? (size_t)2 * p.Cin * sizeof(float) : 0;
    // single K step: one slot each, more workgroups per CU
    const size_t lds = (nk > 1 ? full : (size_t)(BM + BN) * 64 * sizeof(bf16_t)) + aff_bytes;
    auto kern = conv_igemm_kernel<BM, BN, WAVES_M, WAVES_N, NSX, CHUNK_MAJOR>;
    static DynLdsOnce once;
    RFD_TRY(once.ensure(reinterpret_cast<const void *>(kern), (int)(full + 16384)));
    if (note_launch("conv_igemm_kernel<%d, %d, %d, %d, %d, %s>", BM, BN, WAVES_M, WAVES_N, NSX, CHUNK_MAJOR ? "true" : "false")) return RFD_OK;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES_M * WAVES_N * 64), lds, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSX> static int launch_conv_cfg(const ConvParams &p, hipStream_t s)
{
    return p.k_chunk_major ? launch_conv_cfg_order<BM, BN, WAVES_M, WAVES_N, NSX, true>(p, s)
                           : launch_conv_cfg_order<BM, BN, WAVES_M, WAVES_N, NSX, false>(p, s);
}

int launch_conv(const ConvParams &p, hipStream_t s)
{
    // every kernel addresses its tensors through 32-bit buffer descriptors and offsets: inputs, outputs (M x Cout, M x ldy) and,
    // for a back-to-back pair, conv1's output (M x n1) must each stay below 4 GiB -- checked before ANY path is chosen
    // (round-3 advisor finding: the pair branch used to return before the guard; B >= 328 at the stage 1 -> 2 boundary wrapped)
    {
        const size_t Mo = (size_t)p.B * p.Ho * p.Wo, lim = 0xfffffff0ull;
        const size_t in1 = (size_t)p.B * p.H * p.W * (p.ldx ? p.ldx : p.Cin) * 2, in2 = (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2;
        const size_t out = Mo * (size_t)std::max(p.Cout, p.ldy) * (p.yf ? 4 : 2), out1 = p.w1 ? Mo * (size_t)p.n1 * 2 : 0;
        if (in1 >= lim || in2 >= lim || out >= lim || out1 >= lim) {
            set_error("conv: a tensor of %zu bytes exceeds the 4 GiB buffer-addressing limit; lower max_batch_size",
                      std::max(std::max(in1, in2), std::max(out, out1)));
            return RFD_ERR_CAPACITY;
        }
    }
    if (p.w1) {
        // conv3 of a dim-match unit + the next unit's conv1 (OP_B2B beyond stage 1).  One persistent kernel where pw_stream
        // itself would run (force_tile 7 / 1 / 2 and small batches: two launches; bit-identical either way).
        const int M1 = p.B * p.Ho * p.Wo;
        // two shapes: (i) a middle unit of stage 2: 128 -> 512, raw sum out, conv1 512 -> 128 on relu(BN(raw));
        //             (ii) the last unit of stage 1: 64 -> 256, activated output only, conv1 256 -> 128 of stage 2's first unit on it
        const bool act_out = !p.y && p.y2;
        const bool s3 = !act_out && p.Cin == 256 && p.Cout == 1024 && p.y && !p.y2 && p.ldy == p.Cout; // stage 3's middle units: pw_pair_kernel only
        const bool b23 = act_out && p.Cin == 128 && p.Cout == 512 && p.n1 == 256;                     // stage 2 -> 3 boundary: pw_pair_kernel only
        const bool shape = s3 || b23 || (act_out ? (p.Cin == 64 && p.Cout == 256 && p.n1 == 128) : (p.Cin == 128 && p.Cout == 512 && p.n1 == 128 && p.y && !p.y2 && p.ldy == p.Cout));
        const int N1 = p.n1;
        // RFD_PW_PAIR=1: pw_pair_kernel for every pair (A/B against pw_b2b_kernel); default: stage 3 only
        static const int pair_all = [] { const char *e = getenv("RFD_PW_PAIR"); return e ? atoi(e) : 0; }();
        // the first unit of stage 2: conv3 128 -> 512 with the 1x1 stride-2 shortcut 256 -> 512 as second K segment, no residual
        const bool u1 = !act_out && p.Cin == 128 && p.Cin2 == 256 && p.stride2 == 2 && p.Cout == 512 && p.n1 == 128 && !p.res && p.y && !p.y2 && p.ldy == p.Cout && p.bias2;
        const bool fuse = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && (u1 || (shape && p.Cin2 == 0 && p.res)) && !p.in_scale &&
                          !p.res_up2 && !p.res_post && !p.relu && !p.yf && p.ldx == p.Cin && p.x_coff == 0 &&
                          p.y_coff == 0 && p.y_split >= p.Cout && p.n_valid >= p.Cout &&
                          (p.force_tile == 6 || p.force_tile == 16 || (p.force_tile == 0 && M1 >= 128 * 128));
        if (fuse && (const char *)p.w1 > (const char *)p.w && (size_t)((const char *)p.w1 - (const char *)p.w) < (1u << 30)) {
            if (u1) return launch_pw_pair<2, 1, false, 4>(p, s);
            if (s3) {
                if (p.n1 != 256) { set_error("conv pair: stage-3 form instantiated for n1 = 256, got %d", p.n1); return RFD_ERR_INVALID_ARG; }
                return launch_pw_pair<4, 2, false>(p, s);
            }
            if (b23) return launch_pw_pair<2, 2, true>(p, s);
            // the stage 1 -> 2 boundary: both filter banks (96 KiB) resident in LDS, barrier-free (RFD_PW_PAIR=2: the streaming pw_b2b form)
            if (act_out && pair_all != 2) return launch_pw_pair<1, 1, true, 0, 2>(p, s);
            if (pair_all == 1) return act_out ? launch_pw_pair<1, 1, true>(p, s) : launch_pw_pair<2, 1, false>(p, s);
            return act_out ? launch_pw_b2b<1, true>(p, s) : launch_pw_b2b<2, false>(p, s);
        }
        ConvParams a = p;
        a.w1 = nullptr; a.bias1 = nullptr; a.t1 = nullptr;
        RFD_TRY(launch_conv(a, s));
        ConvParams q;
        memset(&q, 0, sizeof q);
        q.x = act_out ? p.y2 : p.y; q.w = p.w1; q.bias = p.bias1; q.zero = p.zero;
        if (!act_out) { q.in_scale = p.scale2; q.in_shift = p.shift2; }
        q.y = p.t1;
        q.B = p.B; q.H = q.Ho = p.Ho; q.W = q.Wo = p.Wo; q.Cin = p.Cout; q.Cout = N1;
        q.KH = q.KW = 1; q.stride = 1; q.pad = 0;
        q.ldx = p.Cout; q.ldy = N1; q.y_split = 1 << 30; q.n_valid = 1 << 30; q.relu = 1;
        q.force_tile = p.force_tile == 16 ? 0 : p.force_tile; q.co_running = p.co_running;
        return launch_conv(q, s);
    }
    if (p.Cin % 64 != 0 || p.Cin2 % 64 != 0 || p.Cout % 32 != 0) {
        set_error("conv: Cin=%d must be a multiple of 64 and Cout=%d of 32", p.Cin, p.Cout);
        return RFD_ERR_INVALID_ARG;
    }
    if (p.in_scale && (p.KH != 1 || p.KW != 1 || p.pad != 0 || p.Cin > 2048)) {
        set_error("conv: the input affine is only defined for un-padded 1x1 convs with Cin <= 2048");
        return RFD_ERR_INVALID_ARG;
    }
    if ((size_t)p.B * p.H * p.W * p.ldx * 2 >= 0xfffffff0ull || (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2 >= 0xfffffff0ull) {
        set_error("conv: an input tensor of %zu bytes exceeds the 4 GiB buffer-addressing limit; lower max_batch_size",
                  (size_t)p.B * p.H * p.W * p.Cin * 2);
        return RFD_ERR_CAPACITY;
    }
    const int M = p.B * p.Ho * p.Wo;
    const int nk = (p.KH * p.KW * p.Cin + p.Cin2) / 64;
    // layer shapes conv3x3_halo_kernel accepts (any size, any force_tile): all their kernels accumulate chunk-major
    const bool halo_shape = p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cin2 == 0 && !p.in_scale && !p.res && !p.y2 &&
                            !p.yf && p.y && p.Ho == p.H && p.Wo == p.W && p.n_valid >= p.Cout && p.Cin % 128 == 0 &&
                            (p.Cout % 128 == 0 || p.Cout == 192) && p.Cout <= 512 && (p.y_split >= p.Cout || p.y_split % 8 == 0) &&
                            (p.W % 16 == 0 || p.W == 40);
    if (halo_shape && !p.k_chunk_major) {
        ConvParams q = p;
        q.k_chunk_major = 1;
        return launch_conv(q, s);
    }
    // force_tile 18: the generic 128 x 128 tile with EIGHT waves (32 x 64 wave tiles, two waves per SIMD from one workgroup): for
    // the small-M layers whose grid gives a CU a single workgroup (A/B; bit-identical, same K order)
    if (p.force_tile == 18 && p.Cout % 128 == 0 && !p.in_scale) return launch_conv_cfg<128, 128, 4, 2, 3>(p, s);
    // force_tile 19: the merged-kx 3x3 kernel in its older four-wave form (64 x 64 wave tiles)
    if (p.force_tile == 19 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cin2 == 0 && !p.in_scale && p.Ho == p.H && p.Wo == p.W &&
        p.W >= 3 && p.Cout % 128 == 0)
        return launch_conv3x3_kx<128, 2, 2>(p, s);
    // wave-specialised loader / consumer ring (kernels_ring.hip): force_tile 17 = wherever the shape allows (tests, A/B)
    if (p.force_tile == 17 && conv_ring_supports(p, nullptr)) return launch_conv_ring(p, s);
    {   // The ring runs the small-M, long-K layers it measured faster on, in its generic form: stage-4 conv1 (2048 -> 512: 27.3 vs
        // 28.8 us per 16 images in isolation) and the stride-2 conv2 of stage 4's first unit (48.6 vs 52.7 us); end to end +1.1 %
        // in two alternating A/B pairs on one box (7 798 / 7 734 vs 7 716 / 7 650 img/s; profiles/r04_ab_ring_env.jsonl).  Every
        // other layer stays with the kernels above (the ring is slower there: DESIGN_AB_RECORD.md round 4).  Bit-identical either
        // way (tests/test_ring_gpu.py).  RFD_CONV_RING=0 switches it off, =2 widens it to every generic-form layer with Cout = 512.
        static const int ring_env = [] { const char *e = getenv("RFD_CONV_RING"); return e ? atoi(e) : 1; }();
        bool kx3 = false;
        if (ring_env >= 1 && p.force_tile == 0 && conv_ring_supports(p, &kx3) && !kx3 && p.B * p.Ho * p.Wo <= 128 * 64 &&
            p.KH * p.KW * p.Cin + p.Cin2 >= (ring_env >= 2 ? 512 : 2048) && p.Cout == 512)
            return launch_conv_ring(p, s);
    }
    // short-K, wide-N pointwise layers with a residual: persistent X-stationary streaming kernel (force_tile 1 / 2 / 5 opt out)
    const bool pw_ok = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.Cin2 == 0 && !p.in_scale && !p.yf && p.res &&
                       !p.res_up2 && !p.res_post && p.ldx == p.Cin && p.x_coff == 0 && p.Cout % 128 == 0 && p.Cout >= 4 * p.Cin &&
                       p.y_coff == 0 && p.y_split >= p.Cout && p.n_valid >= p.Cout && (!p.y || p.ldy == p.Cout) &&
                       (p.Cin == 64 || p.Cin == 128 || p.Cin == 256) && (p.force_tile == 0 || p.force_tile == 6 || p.force_tile == 8 ||
                        (p.force_tile == 10 && p.Cin == 128) || (p.force_tile == 11 && p.Cin == 256));
    // a persistent workgroup walks all N / 128 chunks of its tiles one after the other: below ~half a GPU of tiles (small
    // batches; B = 1: 13 tiles at 40 x 40) the generic kernel's tiles_m x N / 128 independent workgroups are faster
    if (pw_ok && (p.y || p.y2) && ((p.Cout >> 7) & 1) == 0 && p.Cout <= 1024 && (M >= 128 * 128 || p.force_tile == 6)) return p.Cin == 64 ? launch_pw_stream_nk<1>(p, s) : (p.Cin == 128 ? launch_pw_stream_nk<2>(p, s) : launch_pw_stream_nk<4>(p, s));
    // wide pointwise GEMMs that pw_stream does not take (conv3 + fused shortcut of the down-sampling units, stage-4 conv3, and
    // N >= 512 conv1s, where the 256 x 256 tile measured faster than pw_gemm's 256 x 128: 32.8 vs 39.3 us for 1024 -> 512 at
    // 40 x 40 x 16): persistent 256 x 256 tiles (force_tile 12: whatever the size; 1 / 2 / 7 opt out)
    const bool pww_ok = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && !p.in_scale && !p.yf && (p.y || p.y2) &&
                        (!p.res || (!p.res_up2 && !p.res_post)) && p.Cin2 % 64 == 0 && p.Cin + p.Cin2 >= 384 && p.Cout % 256 == 0 &&
                        p.Cout >= 512 && p.Cout <= 2048 && p.y_split >= p.Cout && p.n_valid >= p.Cout && M % 8 == 0 &&
                        (p.force_tile == 0 || p.force_tile == 6 || p.force_tile == 12);
    if (pww_ok && (p.force_tile != 0 || ceil_div(M, 256) * (p.Cout / 256) >= 150)) return launch_pw_wide(p, s);
    // long-K pointwise layers without a residual (conv1 of the units): persistent activation-streaming kernel
    // (force_tile 15: whatever the size; 1 / 2 / 7 opt out)
    const bool pwg_ok = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.Cin2 == 0 && !p.y2 && !p.yf && p.Ho == p.H && p.Wo == p.W &&
                        p.y && p.Cin % 128 == 0 && p.Cin >= 256 && p.Cin <= 2048 && p.Cout % 128 == 0 && p.Cout <= 1024 && p.y_split >= p.Cout &&
                        p.n_valid >= p.Cout && M % 8 == 0 && (p.force_tile == 0 || p.force_tile == 6 || p.force_tile == 15);
    // (with a residual -- the FPN laterals -- only from 400 items: 52 vs 57 us for 512 -> 256 at 80 x 80 x 16, but 27 vs 25 us for the
    //  200 items of 1024 -> 256 at 40 x 40)
    if (pwg_ok && (p.force_tile != 0 || ceil_div(M, 256) * (p.Cout / 128) >= (p.res ? 400 : 150))) return launch_pw_gemm(p, s);
    // 64 -> 64 3x3: filter bank resident in LDS, halo tile staged once for all nine taps (force_tile 1 / 2 / 7 opt out)
    const bool c64_ok = p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cin == 64 && p.Cout == 64 && p.Cin2 == 0 && !p.in_scale &&
                        !p.res && !p.y2 && !p.yf && p.y && p.Ho == p.H && p.Wo == p.W && p.y_split >= 64 && p.n_valid >= 64 &&
                        (p.force_tile == 0 || p.force_tile == 6 || p.force_tile == 9);
    if (c64_ok && (M >= 96 * 256 || p.force_tile == 6)) return launch_conv3x3_c64(p, s); // >= 96 tiles of 16 x 16 pixels
    // Cin >= 128 3x3: persistent halo-tile kernel; work items of 128 / 192 / 256 output channels (TN = 4 / 6 / 8).  Every 3x3
    // kernel accumulates in the same order, so the choice changes no bit of the result.
    // force_tile 13: smallest item, 14: largest item, whatever the size; 1 / 2 / 7 opt out
    const bool halo_ok = halo_shape && (p.force_tile == 0 || p.force_tile == 6 || p.force_tile == 13 || p.force_tile == 14);
    if (halo_ok) {
        const int tiles = p.W == 40 ? p.B * ceil_div(p.H, 6) : p.B * (p.W / 16) * ceil_div(p.H, 16);
        const bool forced = p.force_tile != 0;
        if (p.Cout == 192) {
            if (forced || tiles >= 100) return p.W == 40 ? launch_conv3x3_halo<40, 6, 6>(p, s) : launch_conv3x3_halo<16, 16, 6>(p, s);
        } else {
            // (the 256-channel item is not instantiated for the 40-wide row tile: its per-pixel address registers next to 128
            //  accumulators spill to scratch, and scratch reloads share the DMA counters; 128-channel items measured 5 % slower
            //  there at B = 32 and are what the size rule picks at B = 16 anyway)
            const int items8 = p.Cout % 256 == 0 && p.W != 40 ? tiles * (p.Cout / 256) : 0, items4 = tiles * (p.Cout / 128);
            // 256-channel items read the halo half as often, 128-channel items fill the last round of workgroups better:
            // 256 unless its share of busy CU-rounds is clearly lower (B = 32, 80 x 80: 800 items = 3.1 rounds vs 1600 = 6.25)
            auto fill = [](int n) { return (double)n / (ceil_div(n, 256) * 256); };
            const bool use8 = p.force_tile == 14 ? items8 > 0 : (p.force_tile == 13 ? false : items8 >= 200 && fill(items8) >= fill(items4) - 0.05);
            if (use8) return launch_conv3x3_halo<16, 16, 8>(p, s);
            if (forced || items4 >= 200) return p.W == 40 ? launch_conv3x3_halo<40, 6, 4>(p, s) : launch_conv3x3_halo<16, 16, 4>(p, s);
        }
    }
    const bool kx_ok = p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cin2 == 0 && !p.in_scale && p.Ho == p.H &&
                       p.Wo == p.W && p.W >= 3 && p.force_tile != 1 && p.force_tile != 2;
    // Eight waves (32 x 64 wave tiles) since round 4: the layers that come here have small grids (stage-4 conv2: 200 tiles on 256
    // CUs), so a CU mostly holds ONE workgroup, and with four waves that is one wave per SIMD -- nothing covers its barrier, drain
    // and fragment-read latencies (1 725 cycles per 32-MFMA step).  Two waves per SIMD from the same workgroup: 53.8 -> 46.1 us
    // (stage-4 conv2, 16 images), 73 -> 65 us at 32 images where two four-wave workgroups already shared a CU, 30.5 -> 27.4 us
    // (SSH 80 x 80 context conv); same K order, bit-identical (force_tile 19: the four-wave form).
    static const int tile_waves = [] { const char *e = getenv("RFD_TILE_WAVES"); return e ? atoi(e) : 8; }(); // 4: the older forms (A/B)
    if (kx_ok && p.Cout % 128 == 0) return tile_waves == 4 ? launch_conv3x3_kx<128, 2, 2>(p, s) : launch_conv3x3_kx<128, 4, 2>(p, s);
    // (with BN = 64 the merged-kx kernel measured 7 % slower than the generic 128x64 tile at 3 workgroups / CU)
    // Per-layer tile choice for a chain that has the GPU to itself (unsplit passes, B < 16; tools/tile_sweep.py): the
    // 128x64 tile (3 workgroups per CU, twice the grid) wins by 5-28 % where the 128x128 grid cannot give every CU a
    // workgroup, and by 5-11 % on the 1x1 layers whose FLOPs per byte of activation traffic are far below the machine
    // balance (more loads in flight per CU).  With a second chain co-running (batch split) the other chain already
    // fills those gaps and the same choice measured 2.5 % SLOWER end to end, so it is not applied there.
    const long long n128 = (long long)ceil_div(M, 128) * (p.Cout / 128);
    const double act_bytes = 2.0 * ((double)(p.KH * p.KW * p.Cin + p.Cin2) / (p.stride * p.stride) +
                                    (double)p.Cout * (1 + (p.res ? 1 : 0) + (p.y && p.y2 ? 1 : 0)));
    const double flop_per_byte = 2.0 * (p.KH * p.KW * p.Cin + p.Cin2) * p.Cout / act_bytes;
    const bool prefer_small = p.force_tile == 0 && !p.co_running && (n128 <= 256 || (p.KH == 1 && flop_per_byte < 110.0));
    if (p.Cout % 128 == 0 && p.force_tile != 4 && !prefer_small) {
        // The 8-wave 256x128 tile with a 3-slot ring (1 workgroup per CU) measured 5-13 % SLOWER than two
        // co-resident 128x128 workgroups on every layer of this network (profiles/): opt-in only.
        (void)M; (void)nk;
        if (p.force_tile == 2) return launch_conv_cfg<256, 128, 4, 2, 3>(p, s);
        // the 80 KiB ring leaves no room for the input-affine table next to a second workgroup
        if (p.force_tile == 1 || p.in_scale) return launch_conv_cfg<128, 128, 2, 2, 2>(p, s);
        // eight waves on the 128 x 128 tile (round 4; as in the merged-kx kernel above): 46.5 vs 49.5 us and 39.2 vs 41.0 us on the
        // stride-2 3x3 layers, 24.8 vs 26.2 us on the 1024 -> 256 lateral; the K = 2048 lateral ties (26.5 vs 26.0) and keeps four
        if (tile_waves != 4 && !(p.KH == 1 && p.Cin + p.Cin2 >= 2048)) return launch_conv_cfg<128, 128, 4, 2, 3>(p, s);
        return launch_conv_cfg<128, 128, 2, 2, 3>(p, s);
    }
    // fused SSH pair (conv1 + ctx1 along N): eight waves (32 x 96 wave tiles) since round 4 -- these layers have at most 50 tiles
    // below the halo kernel's threshold, one workgroup per CU, and the four-wave form needs 284 registers (one wave per SIMD)
    if (p.Cout % 192 == 0 && p.Cout % 128 != 0) return (tile_waves == 4 || p.force_tile == 1 || p.force_tile == 19) ? launch_conv_cfg<128, 192, 2, 2, 2>(p, s) : launch_conv_cfg<128, 192, 4, 2, 2>(p, s);
    // 128x64: the 2-slot ring keeps 3 workgroups per CU, which measured faster than a deeper ring at 2
    if (p.Cout % 64 == 0) {
        if (p.force_tile == 3) return launch_conv_cfg<256, 64, 4, 1, 2>(p, s); // 64x64 wave tiles, 2 workgroups / CU
        return launch_conv_cfg<128, 64, 4, 1, 2>(p, s);
    }
    return launch_conv_cfg<128, 32, 4, 1, 2>(p, s);
}

// ------------------------------------------------------------------------------------------------
// conv0: 7x7 stride 2 pad 3, Cin = 3 (+1 zero channel), Cout = 64, fused bias + ReLU.
// K is laid out as 7 (ky) x 32 (8 kx x 4 channels; kx = 7 and channel 3 carry zero weights), so each
// ky is exactly one 16x16x32 MFMA K step and a lane's 8-element B fragment is two adjacent input
// pixels (16 contiguous bytes of the NHWC4 image).  Each wave keeps all 64x224 weights in registers
// (28 A fragments) and streams 16-pixel output tiles.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv0_kernel(const bf16_t *__restrict__ x4,
                                                    const bf16_t *__restrict__ w, // [64][7][32]
                                                    const float *__restrict__ bias,
                                                    bf16_t *__restrict__ y, int B, int H, int W)
{
    const int lane = threadIdx.x & 63, frow = lane & 15, fq = lane >> 4;
    const int Ho = H >> 1, Wo = W >> 1;
    const long long M = (long long)B * Ho * Wo;
    const long long ntile = (M + 15) >> 4;
    bf16x8 af[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 7; ++k)
            af[i][k] = *reinterpret_cast<const bf16x8 *>(w + ((size_t)(i * 16 + frow) * 7 + k) * 32 + fq * 8);
    float4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const float4 *>(bias + i * 16 + fq * 4);

    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long t = wave_id; t < ntile; t += nwaves) {
        const long long m = t * 16 + frow;
        const bool mok = m < M;
        int b = 0, ho = 0, wo = 0;
        if (mok) {
            b = (int)(m / ((long long)Ho * Wo));
            const int rem = (int)(m - (long long)b * Ho * Wo);
            ho = rem / Wo;
            wo = rem - ho * Wo;
        }
        const int wi = 2 * wo - 3 + 2 * fq; // first of this lane's two input pixels
        const bool w0ok = mok && (unsigned)wi < (unsigned)W, w1ok = mok && (unsigned)(wi + 1) < (unsigned)W;
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        uint4 frag[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int hi = 2 * ho - 3 + k;
            const bool hok = (unsigned)hi < (unsigned)H;
            const uint2 *px = reinterpret_cast<const uint2 *>(x4) + ((long long)b * H + hi) * W + wi;
            const uint2 p0 = (hok && w0ok) ? px[0] : make_uint2(0, 0);
            const uint2 p1 = (hok && w1ok) ? px[1] : make_uint2(0, 0);
            frag[k] = make_uint4(p0.x, p0.y, p1.x, p1.y);
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const bf16x8 bf = __builtin_bit_cast(bf16x8, frag[k]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][k], bf, acc[i], 0, 0, 0);
        }
        if (mok) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<uint2 *>(y + m * 64 + i * 16 + fq * 4) =
                    pack_bf16x4(fmaxf(acc[i][0] + bv[i].x, 0.f), fmaxf(acc[i][1] + bv[i].y, 0.f),
                                fmaxf(acc[i][2] + bv[i].z, 0.f), fmaxf(acc[i][3] + bv[i].w, 0.f));
        }
    }
}

int launch_conv0(const bf16_t *x4, const bf16_t *w, const float *bias, bf16_t *y, int B, int H, int W,
                 hipStream_t s)
{
    if ((H | W) & 1) {
        set_error("conv0: input %dx%d must be even", H, W);
        return RFD_ERR_INVALID_ARG;
    }
    const long long ntile = ((long long)B * (H / 2) * (W / 2) + 15) / 16;
    const int grid = (int)std::min<long long>((ntile + 3) / 4, 256 * 8);
    if (note_launch("conv0_kernel")) return RFD_OK;
    hipLaunchKernelGGL(conv0_kernel, dim3(grid), dim3(256), 0, s, x4, w, bias, y, B, H, W);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused stem: conv0 (7x7/2, +bias, ReLU) -> 3x3/2 max pool (pad 1) -> per-channel affine + ReLU, one kernel.
// A workgroup produces a 4 x 16 tile of POOLED pixels: it stages the 23 x 71 input patch in LDS (zero
// padded), its 4 waves compute the 9 x 33 conv pixels the pool windows need (16-pixel MFMA tiles, weights
// resident in registers, B fragments = 16-byte LDS reads of two adjacent input pixels) into an LDS tile,
// and the pooled + activated result leaves as 16-byte stores.  The 13 MB/image conv0 activation never
// reaches HBM (840 MB per 32-image batch saved: one write, one read).
// Out-of-image conv pixels (pool padding) are stored as 0: exact, because every real value is >= 0 (ReLU).
// ------------------------------------------------------------------------------------------------
constexpr int kStemPH = 4, kStemPW = 16;                       // pooled tile
constexpr int kStemCR = 2 * kStemPH + 1, kStemCC = 2 * kStemPW + 1; // conv pixels: 9 x 33
constexpr int kStemIR = 2 * kStemCR + 5, kStemIC = 2 * kStemCC + 5; // input patch: 23 x 71
constexpr int kStemIP = 72;                                    // input row pitch in pixels (8 B each)
constexpr int kStemCP = 72;                                    // conv tile pitch in bf16 per pixel (144 B)

__global__ void __launch_bounds__(256) stem_kernel(const bf16_t *__restrict__ x4, const bf16_t *__restrict__ w,
                                                   const float *__restrict__ bias, const float *__restrict__ scale,
                                                   const float *__restrict__ shift, bf16_t *__restrict__ y, int H,
                                                   int W, int tiles_w, int tiles_h)
{
    __shared__ __attribute__((aligned(16))) uint2 in_tile[kStemIR * kStemIP];
    __shared__ __attribute__((aligned(16))) bf16_t conv_tile[kStemCR * kStemCC * kStemCP];
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fq = lane >> 4;
    // scalar wave index: the loops over `wave` below are then scalar loops in the ISA, i.e. their 128-bit LDS reads provably
    // run with EXEC all ones (tools/isa_check.py, tests/test_build_cpu.py)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ho = H >> 1, Wo = W >> 1, Hp = Ho >> 1, Wp = Wo >> 1;
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h;
    const int b = bid / tiles_h;
    const int ph0 = th * kStemPH, pw0 = tw * kStemPW;
    const int cr0 = 2 * ph0 - 1, cc0 = 2 * pw0 - 1; // first conv pixel
    const int ir0 = 2 * cr0 - 3, ic0 = 2 * cc0 - 3; // first input pixel

    // weights: 64 x [7][32] resident in registers; A-operand row rho = i*16 + frow holds output channel
    // (i>>1)*32 + (frow>>2)*8 + (i&1)*4 + (frow&3), so a lane's accumulators are 8 consecutive channels
    bf16x8 af[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chn = (i >> 1) * 32 + (frow >> 2) * 8 + (i & 1) * 4 + (frow & 3);
#pragma unroll
        for (int k = 0; k < 7; ++k)
            af[i][k] = *reinterpret_cast<const bf16x8 *>(w + ((size_t)chn * 7 + k) * 32 + fq * 8);
    }
    // input patch -> LDS (zero outside the image)
    const uint2 *src = reinterpret_cast<const uint2 *>(x4) + (size_t)b * H * W;
    for (int round = 0; round < (kStemIR * kStemIP + 255) / 256; ++round) { // scalar trip count, predicated body
        const int i = tid + round * 256;
        const int r = i / kStemIP, c = i - r * kStemIP;
        const int gr = ir0 + r, gc = ic0 + c;
        uint2 v = make_uint2(0, 0);
        if (c < kStemIC && (unsigned)gr < (unsigned)H && (unsigned)gc < (unsigned)W) v = src[(size_t)gr * W + gc];
        if (i < kStemIR * kStemIP) in_tile[i] = v;
    }
    __syncthreads();

    float bv[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 b0 = *reinterpret_cast<const float4 *>(bias + h * 32 + fq * 8);
        const float4 b1 = *reinterpret_cast<const float4 *>(bias + h * 32 + fq * 8 + 4);
        bv[h][0] = b0.x; bv[h][1] = b0.y; bv[h][2] = b0.z; bv[h][3] = b0.w;
        bv[h][4] = b1.x; bv[h][5] = b1.y; bv[h][6] = b1.z; bv[h][7] = b1.w;
    }
    constexpr int NPIX = kStemCR * kStemCC, NT16 = (NPIX + 15) / 16;
    for (int t = wave; t < NT16; t += 4) {
        const int idx = min(t * 16 + frow, NPIX - 1);
        const int cr = idx / kStemCC, cc = idx - cr * kStemCC;
        const uint2 *ip = in_tile + (2 * cr) * kStemIP + 2 * cc + 2 * fq; // 16-byte aligned: even pixel index
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(ip + k * kStemIP);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][k], bf, acc[i], 0, 0, 0);
        }
        if (t * 16 + frow < NPIX) {
            const int gr = cr0 + cr, gc = cc0 + cc;
            // out-of-image conv pixels become +0 by masking the packed bits (a select per value compiled into a branch each)
            const uint32_t inside = ((unsigned)gr < (unsigned)Ho && (unsigned)gc < (unsigned)Wo) ? 0xffffffffu : 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = fmaxf(acc[2 * h][k] + bv[h][k], 0.f);
                    o[4 + k] = fmaxf(acc[2 * h + 1][k] + bv[h][4 + k], 0.f);
                }
                const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                *reinterpret_cast<uint4 *>(conv_tile + idx * kStemCP + h * 32 + fq * 8) =
                    make_uint4(lo.x & inside, lo.y & inside, hi.x & inside, hi.y & inside);
            }
        }
    }
    __syncthreads();

    // 3x3/2 max pool over the conv tile, then affine + ReLU; 8 channels (16 bytes) per item
    static_assert(kStemPH * kStemPW * 8 % 256 == 0, "the pooling pass runs whole rounds of the workgroup (a scalar trip count)");
    for (int round = 0; round < kStemPH * kStemPW * 8 / 256; ++round) {
        const int item = tid + round * 256;
        const int c8 = item & 7, pp = item >> 3;
        const int pr = pp / kStemPW, pc = pp - pr * kStemPW;
        // no early `continue`: the ds_read_b128 below must run with EXEC all ones ("a hardware
        // observation": 128-bit LDS reads under a partial EXEC mask return wrong data in lanes 48-63 while MFMA waves of
        // another kernel share the CU); only the store is predicated
        const bool live = ph0 + pr < Hp && pw0 + pc < Wp;
        // The conv tile holds ReLU outputs: non-negative bf16 (or -0), whose bit patterns order like signed 16-bit integers
        // (-0 = 0x8000 is the smallest and loses against the initial +0, as it does in float).  So the 3x3 max runs on packed
        // 16-bit integers, two channels per instruction, and only the result is widened: 36 instead of 144 VALU operations.
        typedef short i16x2 __attribute__((ext_vector_type(2)));
        i16x2 mi[4] = {i16x2{0, 0}, i16x2{0, 0}, i16x2{0, 0}, i16x2{0, 0}};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const uint4 v = *reinterpret_cast<const uint4 *>(conv_tile + ((2 * pr + dy) * kStemCC + 2 * pc + dx) * kStemCP + c8 * 8);
                const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) mi[k] = __builtin_elementwise_max(mi[k], __builtin_bit_cast(i16x2, u[k]));
            }
        // pin the reads above the `if (live)`: hipcc otherwise sinks the whole body, reads included, under the store's predicate
        // (seen in the round-3 ISA audit, tools/isa_check.py) -- a volatile asm cannot move into a conditional block
        asm volatile("" : "+v"(mi[0]), "+v"(mi[1]), "+v"(mi[2]), "+v"(mi[3]));
        float mx[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t u = __builtin_bit_cast(uint32_t, mi[k]);
            mx[2 * k] = bf16_bits_to_f32(u & 0xffffu);
            mx[2 * k + 1] = bf16_bits_to_f32(u >> 16);
        }
        const float4 s0 = *reinterpret_cast<const float4 *>(scale + c8 * 8), s1 = *reinterpret_cast<const float4 *>(scale + c8 * 8 + 4);
        const float4 t0 = *reinterpret_cast<const float4 *>(shift + c8 * 8), t1 = *reinterpret_cast<const float4 *>(shift + c8 * 8 + 4);
        const uint2 lo = pack_bf16x4(fmaxf(mx[0] * s0.x + t0.x, 0.f), fmaxf(mx[1] * s0.y + t0.y, 0.f),
                                     fmaxf(mx[2] * s0.z + t0.z, 0.f), fmaxf(mx[3] * s0.w + t0.w, 0.f));
        const uint2 hi = pack_bf16x4(fmaxf(mx[4] * s1.x + t1.x, 0.f), fmaxf(mx[5] * s1.y + t1.y, 0.f),
                                     fmaxf(mx[6] * s1.z + t1.z, 0.f), fmaxf(mx[7] * s1.w + t1.w, 0.f));
        if (live) *reinterpret_cast<uint4 *>(y + (((size_t)b * Hp + ph0 + pr) * Wp + pw0 + pc) * 64 + c8 * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

// PERSISTENT form (round 4): a workgroup walks a contiguous run of tiles.  The 28 weight fragments and the bias stay in registers
// across tiles (the one-tile kernel re-reads 28 KiB of weights from L2 per 64 pooled pixels), and the NEXT tile's input patch is
// requested into registers before the conv phase and written to LDS behind it, so its latency hides under the MFMAs instead of
// opening every tile.  Same arithmetic per tile: bit-identical output.
// FUSE1 (round 4): the first unit's conv1 (1x1, 64 -> 64, + bias, ReLU) rides on the pooled tile.  The pooling pass maps a lane to
// (pooled pixel = lane & 15 of the wave's pooled row, channel group = lane >> 4 [+ 4 in the second round]) -- the B-fragment layout
// of the 16 x 16 x 32 MFMA -- so the packed bf16 outputs it stores to y ARE conv1's operand; W1 (8 KiB) sits in LDS, 8 MFMAs per
// wave and tile, and t1 = relu(W1 . y + b1) is stored next to y.  Same operand bits, same two K steps on a zero accumulator, same
// epilogue arithmetic as the generic kernel on the stored y: bit-identical, and y is not read back (105 MB per 32 images).
template <bool FUSE1>
__global__ void __launch_bounds__(256) stem_persistent_kernel(const bf16_t *__restrict__ x4, const bf16_t *__restrict__ w,
                                                              const float *__restrict__ bias, const float *__restrict__ scale,
                                                              const float *__restrict__ shift, bf16_t *__restrict__ y, int H,
                                                              int W, int tiles_w, int tiles_h, int ntiles, int per,
                                                              const bf16_t *__restrict__ w1, const float *__restrict__ bias1,
                                                              bf16_t *__restrict__ t1)
{
    RFD_CLOCK(12);
    __shared__ __attribute__((aligned(16))) uint2 in_tile[kStemIR * kStemIP];
    __shared__ __attribute__((aligned(16))) bf16_t conv_tile[kStemCR * kStemCC * kStemCP];
    __shared__ __attribute__((aligned(16))) bf16_t w1_tile[FUSE1 ? 64 * 64 : 8]; // row rho = i*16 + r holds output channel (i>>1)*32 + (r>>2)*8 + (i&1)*4 + (r&3); 16-byte chunk c at (c ^ (rho & 7))
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 15, fq = lane >> 4;
    // scalar wave index: the loops over `wave` below are then scalar loops in the ISA, i.e. their 128-bit LDS reads provably
    // run with EXEC all ones (tools/isa_check.py, tests/test_build_cpu.py)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ho = H >> 1, Wo = W >> 1, Hp = Ho >> 1, Wp = Wo >> 1;
    const int t_begin = (int)blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    if (t_begin >= t_end) return;
    constexpr int NR = (kStemIR * kStemIP + 255) / 256; // patch elements per thread
    // patch of tile t: global loads into registers (zero outside the image); written to LDS by store_patch
    auto load_patch = [&](int t, uint2 (&pre)[NR]) {
        const int tw_ = t % tiles_w, th_ = (t / tiles_w) % tiles_h, b_ = t / (tiles_w * tiles_h);
        const int ir0_ = 2 * (2 * th_ * kStemPH - 1) - 3, ic0_ = 2 * (2 * tw_ * kStemPW - 1) - 3;
        const uint2 *src_ = reinterpret_cast<const uint2 *>(x4) + (size_t)b_ * H * W;
#pragma unroll
        for (int round = 0; round < NR; ++round) {
            const int i = tid + round * 256;
            const int r = i / kStemIP, c = i - r * kStemIP;
            const int gr = ir0_ + r, gc = ic0_ + c;
            // unconditional load from a clamped address, zeroed by a mask: a load under a per-element condition makes hipcc branch
            // around each one and wait for it there (cdna_hip_programming.md section 5, trap 4(c)) -- seven serialised round trips
            const bool ok = c < kStemIC && (unsigned)gr < (unsigned)H && (unsigned)gc < (unsigned)W;
            const uint32_t keep = ok ? 0xffffffffu : 0u;
            const uint2 v = src_[ok ? (size_t)gr * W + gc : (size_t)0];
            pre[round] = make_uint2(v.x & keep, v.y & keep);
        }
    };
    auto store_patch = [&](const uint2 (&pre)[NR]) {
#pragma unroll
        for (int round = 0; round < NR; ++round) {
            const int i = tid + round * 256;
            if (i < kStemIR * kStemIP) in_tile[i] = pre[round];
        }
    };

    // weights: 64 x [7][32] resident in registers; A-operand row rho = i*16 + frow holds output channel
    // (i>>1)*32 + (frow>>2)*8 + (i&1)*4 + (frow&3), so a lane's accumulators are 8 consecutive channels
    bf16x8 af[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int chn = (i >> 1) * 32 + (frow >> 2) * 8 + (i & 1) * 4 + (frow & 3);
#pragma unroll
        for (int k = 0; k < 7; ++k)
            af[i][k] = *reinterpret_cast<const bf16x8 *>(w + ((size_t)chn * 7 + k) * 32 + fq * 8);
    }
    uint2 pre[NR];
    load_patch(t_begin, pre);
    store_patch(pre);
    if (FUSE1) {
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const int idx = tid + round * 256, rho = idx >> 3, c = idx & 7, i = rho >> 4, r = rho & 15;
            const int chn = (i >> 1) * 32 + (r >> 2) * 8 + (i & 1) * 4 + (r & 3);
            *reinterpret_cast<uint4 *>(w1_tile + rho * 64 + ((c ^ (rho & 7)) << 3)) = *reinterpret_cast<const uint4 *>(w1 + chn * 64 + c * 8);
        }
    }
    __syncthreads();

    float bv[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 b0 = *reinterpret_cast<const float4 *>(bias + h * 32 + fq * 8);
        const float4 b1 = *reinterpret_cast<const float4 *>(bias + h * 32 + fq * 8 + 4);
        bv[h][0] = b0.x; bv[h][1] = b0.y; bv[h][2] = b0.z; bv[h][3] = b0.w;
        bv[h][4] = b1.x; bv[h][5] = b1.y; bv[h][6] = b1.z; bv[h][7] = b1.w;
    }
    constexpr int NPIX = kStemCR * kStemCC, NT16 = (NPIX + 15) / 16;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, b = tile / (tiles_w * tiles_h);
        const int ph0 = th * kStemPH, pw0 = tw * kStemPW;
        const int cr0 = 2 * ph0 - 1, cc0 = 2 * pw0 - 1; // first conv pixel
        const bool more = tile + 1 < t_end;               // workgroup-uniform
        if (more) load_patch(tile + 1, pre);              // in flight under the conv phase
    for (int t = wave; t < NT16; t += 4) {
        const int idx = min(t * 16 + frow, NPIX - 1);
        const int cr = idx / kStemCC, cc = idx - cr * kStemCC;
        const uint2 *ip = in_tile + (2 * cr) * kStemIP + 2 * cc + 2 * fq; // 16-byte aligned: even pixel index
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(ip + k * kStemIP);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][k], bf, acc[i], 0, 0, 0);
        }
        if (t * 16 + frow < NPIX) {
            const int gr = cr0 + cr, gc = cc0 + cc;
            // out-of-image conv pixels become +0 by masking the packed bits (a select per value compiled into a branch each)
            const uint32_t inside = ((unsigned)gr < (unsigned)Ho && (unsigned)gc < (unsigned)Wo) ? 0xffffffffu : 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float o[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    o[k] = fmaxf(acc[2 * h][k] + bv[h][k], 0.f);
                    o[4 + k] = fmaxf(acc[2 * h + 1][k] + bv[h][4 + k], 0.f);
                }
                const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
                *reinterpret_cast<uint4 *>(conv_tile + idx * kStemCP + h * 32 + fq * 8) =
                    make_uint4(lo.x & inside, lo.y & inside, hi.x & inside, hi.y & inside);
            }
        }
    }
        __syncthreads();                                  // conv tile complete; nobody reads the input patch any more
        if (more) store_patch(pre);
    // 3x3/2 max pool over the conv tile, then affine + ReLU; 8 channels (16 bytes) per item
    static_assert(kStemPH * kStemPW * 8 % 256 == 0, "the pooling pass runs whole rounds of the workgroup (a scalar trip count)");
    static_assert(!FUSE1 || (kStemPW == 16 && kStemPH == 4), "fused conv1: a wave owns one pooled row of 16 pixels");
    uint4 yq[2]; // FUSE1: this lane's B fragments -- channels (round * 4 + fq) * 8 .. + 7 of pooled pixel (wave, frow)
    for (int round = 0; round < kStemPH * kStemPW * 8 / 256; ++round) {
        const int item = tid + round * 256;
        const int c8 = FUSE1 ? round * 4 + fq : item & 7, pp = FUSE1 ? wave * 16 + frow : item >> 3;
        const int pr = pp / kStemPW, pc = pp - pr * kStemPW;
        // no early `continue`: the ds_read_b128 below must run with EXEC all ones ("a hardware
        // observation": 128-bit LDS reads under a partial EXEC mask return wrong data in lanes 48-63 while MFMA waves of
        // another kernel share the CU); only the store is predicated
        const bool live = ph0 + pr < Hp && pw0 + pc < Wp;
        // The conv tile holds ReLU outputs: non-negative bf16 (or -0), whose bit patterns order like signed 16-bit integers
        // (-0 = 0x8000 is the smallest and loses against the initial +0, as it does in float).  So the 3x3 max runs on packed
        // 16-bit integers, two channels per instruction, and only the result is widened: 36 instead of 144 VALU operations.
        typedef short i16x2 __attribute__((ext_vector_type(2)));
        i16x2 mi[4] = {i16x2{0, 0}, i16x2{0, 0}, i16x2{0, 0}, i16x2{0, 0}};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const uint4 v = *reinterpret_cast<const uint4 *>(conv_tile + ((2 * pr + dy) * kStemCC + 2 * pc + dx) * kStemCP + c8 * 8);
                const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) mi[k] = __builtin_elementwise_max(mi[k], __builtin_bit_cast(i16x2, u[k]));
            }
        // pin the reads above the `if (live)`: hipcc otherwise sinks the whole body, reads included, under the store's predicate
        // (seen in the round-3 ISA audit, tools/isa_check.py) -- a volatile asm cannot move into a conditional block
        asm volatile("" : "+v"(mi[0]), "+v"(mi[1]), "+v"(mi[2]), "+v"(mi[3]));
        float mx[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t u = __builtin_bit_cast(uint32_t, mi[k]);
            mx[2 * k] = bf16_bits_to_f32(u & 0xffffu);
            mx[2 * k + 1] = bf16_bits_to_f32(u >> 16);
        }
        const float4 s0 = *reinterpret_cast<const float4 *>(scale + c8 * 8), s1 = *reinterpret_cast<const float4 *>(scale + c8 * 8 + 4);
        const float4 t0 = *reinterpret_cast<const float4 *>(shift + c8 * 8), t1 = *reinterpret_cast<const float4 *>(shift + c8 * 8 + 4);
        const uint2 lo = pack_bf16x4(fmaxf(mx[0] * s0.x + t0.x, 0.f), fmaxf(mx[1] * s0.y + t0.y, 0.f),
                                     fmaxf(mx[2] * s0.z + t0.z, 0.f), fmaxf(mx[3] * s0.w + t0.w, 0.f));
        const uint2 hi = pack_bf16x4(fmaxf(mx[4] * s1.x + t1.x, 0.f), fmaxf(mx[5] * s1.y + t1.y, 0.f),
                                     fmaxf(mx[6] * s1.z + t1.z, 0.f), fmaxf(mx[7] * s1.w + t1.w, 0.f));
        if (live) *reinterpret_cast<uint4 *>(y + (((size_t)b * Hp + ph0 + pr) * Wp + pw0 + pc) * 64 + c8 * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (FUSE1) yq[round & 1] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    if (FUSE1) { // conv1 on the pooled row of this wave: t1[pixel][n] = relu(sum_k W1[n][k] y[pixel][k] + b1[n])
        f32x4 a1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 bq = __builtin_bit_cast(bf16x8, yq[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rho = i * 16 + frow;
                const bf16x8 aw = *reinterpret_cast<const bf16x8 *>(w1_tile + rho * 64 + (((kk * 4 + fq) ^ (rho & 7)) << 3));
                a1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw, bq, a1[i], 0, 0, 0);
            }
        }
        const bool live1 = ph0 + wave < Hp && pw0 + frow < Wp;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 c0 = *reinterpret_cast<const float4 *>(bias1 + h * 32 + fq * 8), c1 = *reinterpret_cast<const float4 *>(bias1 + h * 32 + fq * 8 + 4);
            const float b1v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            float o[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = fmaxf(a1[2 * h][k] + b1v[k], 0.f);
                o[4 + k] = fmaxf(a1[2 * h + 1][k] + b1v[4 + k], 0.f);
            }
            const uint2 lo = pack_bf16x4(o[0], o[1], o[2], o[3]), hi = pack_bf16x4(o[4], o[5], o[6], o[7]);
            if (live1) *reinterpret_cast<uint4 *>(t1 + (((size_t)b * Hp + ph0 + wave) * Wp + pw0 + frow) * 64 + h * 32 + fq * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
        __syncthreads();                                  // next patch in LDS, conv tile free
    }
}


int launch_stem(const bf16_t *x4, const bf16_t *w, const float *bias, const float *scale, const float *shift,
                bf16_t *y, int B, int H, int W, hipStream_t s, const bf16_t *w1, const float *bias1, bf16_t *t1, bool *fused)
{
    if (fused) *fused = false;
    if ((H | W) & 3) {
        set_error("stem: input %dx%d must be a multiple of 4", H, W);
        return RFD_ERR_INVALID_ARG;
    }
    const int Hp = H / 4, Wp = W / 4;
    const int tiles_h = ceil_div(Hp, kStemPH), tiles_w = ceil_div(Wp, kStemPW);
    const int ntiles = B * tiles_h * tiles_w;
    // persistent form from 4 tiles per workgroup slot (2 workgroups per CU by LDS): weights stay in registers, the next patch is
    // prefetched (RFD_STEM_PERSIST=0: the one-tile kernel, for A/B; bit-identical)
    static const int persist_env = [] { const char *e = getenv("RFD_STEM_PERSIST"); return e ? atoi(e) : 1; }();
    const int slots = 2 * device_cus();
    if (persist_env && ntiles >= 4 * slots) {
        const int per = ceil_div(ntiles, slots), grid = ceil_div(ntiles, per);
        if (w1 && t1 && fused) { // the first unit's conv1 on the pooled tile (Network::run offers it when the next op is that conv)
            *fused = true;
            if (note_launch("stem_persistent_kernel<true>")) return RFD_OK;
            hipLaunchKernelGGL(stem_persistent_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, x4, w, bias, scale, shift, y, H, W, tiles_w, tiles_h, ntiles, per, w1, bias1, t1);
            RFD_HIP(hipGetLastError());
            return RFD_OK;
        }
        if (note_launch("stem_persistent_kernel<false>")) return RFD_OK;
        hipLaunchKernelGGL(stem_persistent_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, x4, w, bias, scale, shift, y, H, W, tiles_w, tiles_h, ntiles, per, nullptr, nullptr, nullptr);
        RFD_HIP(hipGetLastError());
        return RFD_OK;
    }
    if (note_launch("stem_kernel")) return RFD_OK;
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, x4, w, bias, scale, shift, y,
                       H, W, tiles_w, tiles_h);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-2 pad-1 max pool (NHWC bf16), optional fused per-channel affine + ReLU on the output
// (the BN1+ReLU that opens the first pre-activation unit).  8 channels (16 bytes) per thread.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_kernel(const bf16_t *__restrict__ x, bf16_t *__restrict__ y,
                                                      const float *__restrict__ scale,
                                                      const float *__restrict__ shift, int B, int H, int W,
                                                      int C, int Ho, int Wo)
{
    const int cg = C >> 3;
    const long long total = (long long)B * Ho * Wo * cg;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg) * 8;
    long long pix = i / cg;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    float mx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) mx[k] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int hi = 2 * ho - 1 + dy;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int wi = 2 * wo - 1 + dx;
            if ((unsigned)wi >= (unsigned)W) continue;
            const uint4 v = *reinterpret_cast<const uint4 *>(x + (((long long)b * H + hi) * W + wi) * C + c);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mx[2 * k] = fmaxf(mx[2 * k], bf16_bits_to_f32(u[k] & 0xffffu));
                mx[2 * k + 1] = fmaxf(mx[2 * k + 1], bf16_bits_to_f32(u[k] >> 16));
            }
        }
    }
    if (scale) {
#pragma unroll
        for (int k = 0; k < 8; ++k) mx[k] = fmaxf(mx[k] * scale[c + k] + shift[c + k], 0.f);
    }
    const uint2 lo = pack_bf16x4(mx[0], mx[1], mx[2], mx[3]), hi2 = pack_bf16x4(mx[4], mx[5], mx[6], mx[7]);
    *reinterpret_cast<uint4 *>(y + i * 8) = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
}

int launch_maxpool3x3s2(const bf16_t *x, bf16_t *y, const float *scale, const float *shift, int B, int H,
                        int W, int C, hipStream_t s)
{
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    if (note_launch("maxpool_kernel")) return RFD_OK;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, scale,
                       shift, B, H, W, C, Ho, Wo);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

// ------------------------------------------------------------------------------------------------
// MobileNet-0.25 backbone helpers (memory-bound byte work, no MFMA):
//   first3x3_kernel : 3x3 stride-2 pad-1 conv on the NHWC4 input, 3 -> 8 channels (+bias, ReLU); the output
//                     tensor is channel-padded to 64 (zeros) so that every later 1x1/3x3 conv can run on the
//                     MFMA implicit-GEMM kernel with K a multiple of 64.
//   dwconv3x3_kernel: depthwise 3x3 (stride 1 or 2, pad 1) + bias + ReLU, NHWC, 8 channels (16 B) per thread,
//                     weights [9][C] bf16.
// ------------------------------------------------------------------------------------------------
// The 288 weights are wave-uniform and indexed by compile-time constants, so they are fetched with scalar loads
// (s_load_dwordxN of bf16 pairs) straight into SGPRs -- no LDS table.  (An earlier version kept them as floats in LDS
// and read them with broadcast ds_read_b128; on MI355X those reads returned wrong data in lanes 48..63 whenever
// MFMA-issuing waves of ANOTHER kernel (a conv of another stream) shared the CU -- see DESIGN.md, "concurrency".)
__global__ void __launch_bounds__(256) first3x3_kernel(const bf16_t *__restrict__ x4, const uint32_t *__restrict__ w2, // [8][3][3][4] bf16 as pairs
                                                       const float *__restrict__ bias, bf16_t *__restrict__ y, int B,
                                                       int H, int W, int Ho, int Wo, int Cd)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * Ho * Wo) return;
    const int wo = (int)(i % Wo);
    const int ho = (int)((i / Wo) % Ho);
    const int b = (int)(i / ((long long)Wo * Ho));
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = bias[c];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hi = 2 * ho - 1 + ky;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int wi = 2 * wo - 1 + kx;
            if ((unsigned)wi >= (unsigned)W) continue;
            const uint2 p = reinterpret_cast<const uint2 *>(x4)[((long long)b * H + hi) * W + wi];
            const float r = bf16_bits_to_f32(p.x & 0xffffu), g = bf16_bits_to_f32(p.x >> 16), bl = bf16_bits_to_f32(p.y & 0xffffu);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t d0 = w2[c * 18 + (ky * 3 + kx) * 2], d1 = w2[c * 18 + (ky * 3 + kx) * 2 + 1];
                acc[c] += r * bf16_bits_to_f32(d0 & 0xffffu) + g * bf16_bits_to_f32(d0 >> 16) + bl * bf16_bits_to_f32(d1 & 0xffffu);
            }
        }
    }
    uint4 *dst = reinterpret_cast<uint4 *>(y + i * Cd);
    const uint2 lo = pack_bf16x4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    const uint2 hi2 = pack_bf16x4(fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f));
    dst[0] = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
    for (int k = 1; k < Cd / 8; ++k) dst[k] = make_uint4(0, 0, 0, 0);
}

int launch_first3x3(const bf16_t *x4, const bf16_t *w, const float *bias, bf16_t *y, int B, int H, int W, int Cd,
                    hipStream_t s)
{
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * Ho * Wo;
    if (note_launch("first3x3_kernel")) return RFD_OK;
    hipLaunchKernelGGL(first3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x4,
                       reinterpret_cast<const uint32_t *>(w), bias, y, B, H, W, Ho, Wo, Cd);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

__global__ void __launch_bounds__(256) dwconv3x3_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ w, // [9][C]
                                                        const float *__restrict__ bias, bf16_t *__restrict__ y, int B,
                                                        int H, int W, int C, int Ho, int Wo, int stride)
{
    const int cg = C >> 3;
    const long long total = (long long)B * Ho * Wo * cg;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % cg) * 8;
    long long pix = i / cg;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    float acc[8];
    {
        const float4 b0 = *reinterpret_cast<const float4 *>(bias + c), b1 = *reinterpret_cast<const float4 *>(bias + c + 4);
        acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int hi = ho * stride - 1 + ky;
        if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int wi = wo * stride - 1 + kx;
            if ((unsigned)wi >= (unsigned)W) continue;
            const uint4 v = *reinterpret_cast<const uint4 *>(x + (((long long)b * H + hi) * W + wi) * C + c);
            const uint4 k = *reinterpret_cast<const uint4 *>(w + (ky * 3 + kx) * C + c);
            const uint32_t vu[4] = {v.x, v.y, v.z, v.w}, ku[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += bf16_bits_to_f32(vu[e] & 0xffffu) * bf16_bits_to_f32(ku[e] & 0xffffu);
                acc[2 * e + 1] += bf16_bits_to_f32(vu[e] >> 16) * bf16_bits_to_f32(ku[e] >> 16);
            }
        }
    }
    const uint2 lo = pack_bf16x4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    const uint2 hi2 = pack_bf16x4(fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f));
    *reinterpret_cast<uint4 *>(y + i * 8) = make_uint4(lo.x, lo.y, hi2.x, hi2.y);
}

int launch_dwconv3x3(const bf16_t *x, const bf16_t *w, const float *bias, bf16_t *y, int B, int H, int W, int C,
                     int stride, hipStream_t s)
{
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long long total = (long long)B * Ho * Wo * (C / 8);
    if (note_launch("dwconv3x3_kernel")) return RFD_OK;
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, bias, y, B, H, W,
                       C, Ho, Wo, stride);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

// head tensor [B][h][w][32] f32 -> the reference's three NCHW tensors of one level
__global__ void __launch_bounds__(256) heads_to_nchw_kernel(const float *__restrict__ h32, float *__restrict__ cls,
                                                            float *__restrict__ bbox, float *__restrict__ lmk,
                                                            int B, int hw)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x; // over B*32*hw, pos fastest
    if (i >= (long long)B * 32 * hw) return;
    const int pos = (int)(i % hw);
    const int c = (int)((i / hw) % 32);
    const int b = (int)(i / ((long long)hw * 32));
    const float v = h32[((size_t)b * hw + pos) * 32 + c];
    if (c < 4) cls[((size_t)b * 4 + c) * hw + pos] = v;
    else if (c < 12) bbox[((size_t)b * 8 + (c - 4)) * hw + pos] = v;
    else lmk[((size_t)b * 20 + (c - 12)) * hw + pos] = v;
}

int launch_heads_to_nchw(const float *h32, float *cls, float *bbox, float *lmk, int B, int fh, int fw,
                         hipStream_t s)
{
    const long long total = (long long)B * 32 * fh * fw;
    hipLaunchKernelGGL(heads_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h32, cls,
                       bbox, lmk, B, fh * fw);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

} // namespace rfd

// kernels_ring.hip -- wave-specialised implicit-GEMM convolution for the layers whose grid cannot hide memory latency with
// occupancy (small M, long K: stage 4, the FPN / SSH 3x3 layers at 20 x 20 and 40 x 40).
//
// Replaces, for those layers, conv_igemm_kernel / conv3x3_kx_kernel of kernels_conv.hip (the network itself replaces the
// reference's remote forward pass, face_detection.rs:279 model_infer).  Same implicit GEMM, same 128 x 128 output tile, same
// K order per layer and the same MFMA sequence per output element, so the results are bit-identical to those kernels; what
// changes is who waits for what:
//
//   * kernels_conv.hip's tiles drain the vector-memory counter (`s_waitcnt vmcnt(0)`) and pass a workgroup barrier at
//     EVERY K step, with every wave both loading and computing; one workgroup per CU on these small-M layers, so drain, barrier,
//     DMA issue, fragment reads and MFMAs follow one another: 1 500-1 700 cycles per step for 512 cycles of MFMA work.
//   * here a workgroup is 4 LOADER waves + 4 CONSUMER waves (one of each per SIMD: waves w and w + 4 share a SIMD,
//     tools/wave_placement.hip).  Loader waves issue nothing but LDS-DMA (`buffer_load ... lds`) into a ring of K-step slots and
//     publish a slot behind a COUNTED `s_waitcnt vmcnt(N)` that leaves the next D steps in flight (sound: operations retire in
//     issue order per wave, and a loader wave never stores -- tools/isa_check.py enforces it).  Consumer waves wait for a slot's
//     FULL count, read its fragments, bump its FREE count (right behind the reads: LDS executes a wave's instructions in order)
//     and run the MFMAs.  No workgroup barrier in the K loop.
//   What it buys is bounded (DESIGN_AB_RECORD.md, round 4): a CU's load path takes 95-125 GB/s from L2 even with ONE step in flight
//   (tools/ring_fill_bench.hip) -- the round-3 review's Little's-law reading of these layers did not hold -- and a 128 x 128 x 64
//   step moves 32 KiB through that 64-B/clk path for 512 cycles of MFMA work, so the tile is at best half of either roof.  The
//   ring is 5-10 % faster than the barrier kernels on stage 4's 1x1 and stride-2 layers and slower everywhere else; launch_conv
//   routes exactly those layers here.
//
// Two forms of one template:
//   generic  one activation tile (128 rows x 128 B) + one weight tile per K step; ring of 4 slots x 32 KiB, D = 1: one step in
//            flight behind the one being published, two slots of slack (a loader that fills the LAST free slot waits for a whole
//            publish -> poll -> read -> FREE -> poll round trip before it can issue again: 0.35 us per step, measured).
//   KX3      3x3 / stride 1 / pad 1: the three kx taps of a (chunk, ky) share ONE extended activation tile (160 rows) as in
//            conv3x3_kx_kernel; activation ring 3 x 20 KiB, weight ring 6 x 16 KiB, D = 2.
// Flags live in LDS as monotonic counters: FULL[slot] += 1 per loader wave per fill, FREE[slot] += 1 per consumer wave per
// use; fill r of a slot waits for FREE >= 4 r, use r waits for FULL >= 4 (r + 1).  Every spin is bounded: a wave that gives
// up marks an LDS word, the consumer waves copy it to *p.fail (read back by the host as RFD_ERR_HIP -- wrong detections are never
// returned), and it stops waiting.
#include <type_traits>

#include "conv_device.h"

namespace rfd {

#ifndef RFD_RING_EXP
#define RFD_RING_EXP 0 // timing experiments (tools/build_variant.sh; results are garbage): 1 loaders ignore FREE, 2 consumers ignore
#endif                 // FULL, 3 both, 4 loaders publish without waiting for the data, 5 polls without s_sleep; on top of 3 (no
                       // handshake at all): 6 consumers skip the fragment reads, 7 consumers skip the MFMAs, 8 consumers do nothing
#define RFD_RING_NOSYNC (RFD_RING_EXP == 3 || RFD_RING_EXP >= 6)
#ifndef RFD_RING_ROLES
#define RFD_RING_ROLES 0 // how roles are dealt to the 8 waves: 0 = waves 0-3 load, 4-7 compute; 1 = even waves load, odd waves compute
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRingSpinLimit = 1 << 20; // polls (>= 64 cycles each) before a wave gives up: ~30 ms, never expected

__device__ __forceinline__ uint32_t lds_u32(const void *p) { return (uint32_t)(uintptr_t)p; }

// Flag accesses are inline asm on purpose: hipcc treats a pending LDS-DMA as a pending LDS write and would put
// `s_waitcnt vmcnt(0)` in front of any LDS read it can see in a loader wave -- the full drain this kernel exists to avoid.
__device__ __forceinline__ uint32_t ring_flag_read(uint32_t addr) // all lanes read one word; returned as a scalar
{
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void ring_flag_add(uint32_t addr, int lane)
{
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1u) : "memory");
}
// Bounded wait.  A wave that gives up marks the workgroup's GAVE-UP word in LDS and stops waiting; the consumer waves copy
// that word to *p.fail before their epilogue.  (Not a global store here: loader waves must not issue vector stores -- their
// counted vmcnt waits rest on every outstanding operation being an older or younger LDS-DMA.  A loader can only give up
// before the fill it then issues unprotected, and the consumers read that fill before they finish, so they see the mark.)
__device__ __forceinline__ void ring_wait(uint32_t addr, uint32_t target, bool &dead, uint32_t gave_up_word)
{
    if (dead) return;
    for (int it = 0; it < kRingSpinLimit; ++it) {
        if (ring_flag_read(addr) >= target) return;
        if (RFD_RING_EXP != 5) __builtin_amdgcn_s_sleep(1);
    }
    dead = true;
    asm volatile("ds_write_b32 %0, %1" : : "v"(gave_up_word), "v"(1u) : "memory");
}
template <int N> __device__ __forceinline__ void ring_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    if (RFD_RING_EXP == 4 && N != 0) return;
    asm volatile("s_waitcnt vmcnt(%0)" : : "i"(N) : "memory");
}

template <bool KX3> struct RingGeom {
    static constexpr int G = KX3 ? 3 : 1;        // weight steps per activation tile
    static constexpr int XR = KX3 ? 160 : 128;   // rows of an activation slot
    static constexpr int XP = XR / 8 / 4;        // activation pieces (8 rows x 128 B) per loader wave and tile
    static constexpr int WP = 4;                 // weight pieces per loader wave and step (128 rows)
    static constexpr int NSX = KX3 ? 3 : 4;      // activation slots
    static constexpr int NSW = KX3 ? 6 : 4;      // weight slots
#ifndef RFD_RING_D
#define RFD_RING_D 1
#endif
#ifndef RFD_RING_D3
#define RFD_RING_D3 2
#endif
    // Weight steps left in flight behind the one being published.  NOT ring size - 1: a loader that fills the last free slot
    // must then wait for publish -> consumer poll -> fragment reads -> FREE -> its own poll (four LDS round trips, ~0.35 us)
    // before it can issue again, every step; with two or three slots of slack it never waits for a consumer.  L2-served fills
    // land in 250-400 cycles (MI355X_MICROARCH.md), so one or two steps in flight cover the latency.
    static constexpr int D = KX3 ? RFD_RING_D3 : RFD_RING_D;
    static constexpr int NFLAGS = 2 * (NSX + NSW);
    static_assert(NFLAGS <= 31, "flag words 0..30, GAVE-UP word 31");
    static constexpr size_t kSlotBytes = (size_t)(NSX * XR + NSW * 128) * 128;
    // DMAs this wave issued AFTER weight step s - D, at the point where step s (tap kx) has just been issued
    static constexpr int later(int kx)
    {
        int n = WP * D;
        for (int j = 0; j < D; ++j)
            if (KX3 ? ((kx - j) % 3 + 3) % 3 == 0 : true) n += XP; // an activation tile goes out right before weight step 3 g
        return n;
    }
};

// grid: tiles_m x tiles_n workgroups of 512 threads (waves 0-3 load, 4-7 compute); dynamic LDS: see launch_conv_ring
template <bool KX3, bool CHUNK_MAJOR>
__global__ void __launch_bounds__(512) conv_ring_kernel(const ConvParams p)
{
    using R = RingGeom<KX3>;
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, TM = 4, TN = 4;
    constexpr int NSX = R::NSX, NSW = R::NSW, XR = R::XR, XP = R::XP, WP = R::WP, D = R::D, G = R::G;
    static_assert(NSW >= D + 1 && NSX * G >= D + 1, "the ring must hold the steps in flight plus the one being consumed");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t *Xs = reinterpret_cast<bf16_t *>(smem);                       // [NSX][XR*64]
    bf16_t *Ws = Xs + NSX * XR * 64;                                     // [NSW][BN*64]
    uint32_t *Flags = reinterpret_cast<uint32_t *>(Ws + NSW * BN * 64);  // fullX[NSX] freeX[NSX] fullW[NSW] freeW[NSW]
    float *Sc = reinterpret_cast<float *>(Flags + 32);                   // optional input-affine table (generic form)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    // role and index within the role (one loader and one consumer per SIMD: tools/wave_placement.hip)
    const bool is_loader = RFD_RING_ROLES == 0 ? wave8 < 4 : (wave8 & 1) == 0;
    const int wave = RFD_RING_ROLES == 0 ? (wave8 & 3) : (wave8 >> 1);
    const int HoWo = p.Ho * p.Wo;
    const int M = p.B * HoWo;
    const int K1 = p.KH * p.KW * p.Cin;
    const int K = K1 + p.Cin2;
    const int nk1 = K1 >> 6, nsteps = K >> 6;
    const int tiles_n = p.Cout / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;

    if (tid < 32) Flags[tid] = 0;
    if (!KX3 && p.in_scale) {
        for (int c = tid; c < K1; c += 512) {
            Sc[c] = p.in_scale[c];
            Sc[K1 + c] = p.in_shift[c];
        }
    }
    __syncthreads(); // the only workgroup barrier

    const uint32_t fullX = lds_u32(Flags), freeX = fullX + 4 * NSX, fullW = freeX + 4 * NSX, freeW = fullW + 4 * NSW;
    const uint32_t gave_up = fullX + 4 * 31; // Flags[31]
    bool dead = false;

    if (is_loader) {
        // =========================================== loader waves ===========================================
        const int lr = lane >> 3, chunk = (lane & 7) ^ lr;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t *>(p.x), 0, (uint32_t)((size_t)p.B * p.H * p.W * p.ldx * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t *>(p.w), 0, (uint32_t)((size_t)p.Cout * K * 2), 0x00020000);
        // weight pieces: LDS row rho = i*16 + fq*4 + r (the MFMA A-operand row) holds output channel
        // (i>>1)*32 + fq*8 + (i&1)*4 + r of a consumer wave's 64-wide slice (conv_device.h: conv_epilogue's channel map)
        uint32_t woff[WP];
#pragma unroll
        for (int q = 0; q < WP; ++q) {
            const int rho = (wave + 4 * q) * 8 + lr;
            const int rw_ = rho % WN, i_ = rw_ >> 4, fq_ = (rw_ >> 2) & 3, r_ = rw_ & 3;
            const int chn = (rho - rw_) + (i_ >> 1) * 32 + fq_ * 8 + (i_ & 1) * 4 + r_;
            woff[q] = (uint32_t)(((size_t)(n0 + chn) * K + chunk * 8) * 2);
        }
        if constexpr (KX3) {
            // ---- extended-tile row rho = piece*8 + lr holds pixel m0 - 1 + rho at input row y + ky - 1 (conv3x3_kx_kernel) ----
            const int HW = p.H * p.W;
            uint32_t xoff[XP];
            int y0[XP];
#pragma unroll
            for (int q = 0; q < XP; ++q) {
                const int pix = m0 - 1 + (wave + 4 * q) * 8 + lr;
                y0[q] = -(1 << 28);
                xoff[q] = 0;
                if (pix >= 0 && pix < M) {
                    const int b = pix / HW, rem = pix - b * HW;
                    const int y = rem / p.W, x = rem - y * p.W;
                    y0[q] = y - 1;
                    xoff[q] = (uint32_t)(((((long long)b * p.H + y - 1) * p.W + x) * p.ldx + p.x_coff + chunk * 8) * 2);
                }
            }
            const int kc_n = p.Cin >> 6, ngroups = 3 * kc_n;
            int ky = 0, kc = 0, s = 0;
            int xslot = 0, wslot = 0, pxslot = 0, pwslot = 0;
            uint32_t xneed = 0, wneed = 0; // FREE count a slot must have reached before its next fill
            // one weight step (tap kx of the current (chunk, ky)); kx is a compile-time constant: the counted wait's immediate depends on it
            auto step = [&](auto kx_c) __attribute__((always_inline)) {
                constexpr int kx = decltype(kx_c)::value;
                if (kx == 0) {
                    if (xneed && !(RFD_RING_EXP == 1 || RFD_RING_NOSYNC)) ring_wait(freeX + 4 * xslot, xneed, dead, gave_up);
                    const uint32_t rowoff = (uint32_t)(ky * p.W * p.ldx * 2);
#pragma unroll
                    for (int q = 0; q < XP; ++q) {
                        const bool ok = (unsigned)(y0[q] + ky) < (unsigned)p.H;
                        blds16(rx, ok ? xoff[q] + rowoff : kOob, (uint32_t)(kc << 7), Xs + xslot * XR * 64 + (wave + 4 * q) * 512);
                    }
                    if (++xslot == NSX) { xslot = 0; xneed += 4; }
                }
                if (wneed && !(RFD_RING_EXP == 1 || RFD_RING_NOSYNC)) ring_wait(freeW + 4 * wslot, wneed, dead, gave_up);
                const uint32_t col = (uint32_t)((((ky * 3 + kx) * p.Cin) + (kc << 6)) * 2);
#pragma unroll
                for (int q = 0; q < WP; ++q) blds16(rw, woff[q], col, Ws + wslot * BN * 64 + (wave + 4 * q) * 512);
                if (++wslot == NSW) { wslot = 0; wneed += 4; }
                if (s >= D) {
                    // weight step s - D (and everything issued before it) has landed for this wave's pieces
                    ring_vmcnt<R::later(kx)>();
                    constexpr int pkx = ((kx - D) % 3 + 3) % 3;
                    if (pkx == 0) { ring_flag_add(fullX + 4 * pxslot, lane); if (++pxslot == NSX) pxslot = 0; }
                    ring_flag_add(fullW + 4 * pwslot, lane);
                    if (++pwslot == NSW) pwslot = 0;
                }
                ++s;
            };
            for (int g = 0; g < ngroups; ++g) {
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});
                if (++ky == 3) { ky = 0; ++kc; } // chunk-major: the order conv3x3_halo_kernel / conv3x3_kx_kernel are bound to
            }
            ring_vmcnt<0>();
            for (int sp = (s > D ? s - D : 0); sp < s; ++sp) {
                if (sp % 3 == 0) { ring_flag_add(fullX + 4 * pxslot, lane); if (++pxslot == NSX) pxslot = 0; }
                ring_flag_add(fullW + 4 * pwslot, lane);
                if (++pwslot == NSW) pwslot = 0;
            }
        } else {
            // ---- generic implicit GEMM: per-lane im2col bookkeeping as in conv_igemm_kernel ----
            const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<bf16_t *>(p.Cin2 ? p.x2 : p.x), 0,
                (uint32_t)(p.Cin2 ? (size_t)p.B * p.H2 * p.W2 * p.Cin2 * 2 : 0), 0x00020000);
            uint32_t xoff[XP], xoff2[XP];
            int hi0[XP], wi0[XP];
#pragma unroll
            for (int q = 0; q < XP; ++q) {
                const int m = m0 + (wave + 4 * q) * 8 + lr;
                xoff2[q] = kOob;
                if (m < M) {
                    const int b = m / HoWo, rem = m - b * HoWo;
                    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                    hi0[q] = ho * p.stride - p.pad;
                    wi0[q] = wo * p.stride - p.pad;
                    xoff[q] = (uint32_t)(((((long long)b * p.H + hi0[q]) * p.W + wi0[q]) * p.ldx + p.x_coff + chunk * 8) * 2);
                    if (p.Cin2)
                        xoff2[q] = (uint32_t)(((((long long)b * p.H2 + ho * p.stride2) * p.W2 + wo * p.stride2) * p.Cin2 + chunk * 8) * 2);
                } else {
                    hi0[q] = -(1 << 28);
                    wi0[q] = 0;
                    xoff[q] = 0;
                }
            }
            const int kc_n = p.Cin >> 6;
            int ky = 0, kx = 0, kc = 0, wtap = 0, wkc = 0;
            int slot = 0, pslot = 0;
            uint32_t need = 0;
            for (int s = 0; s < nsteps; ++s) {
                if (need && !(RFD_RING_EXP == 1 || RFD_RING_NOSYNC)) ring_wait(freeW + 4 * slot, need, dead, gave_up);
                if (s < nk1) {
                    const uint32_t tap = (uint32_t)((ky * p.W + kx) * p.ldx * 2);
                    const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(kc << 7);
#pragma unroll
                    for (int q = 0; q < XP; ++q) {
                        const bool ok = (unsigned)(hi0[q] + ky) < (unsigned)p.H && (unsigned)(wi0[q] + kx) < (unsigned)p.W;
                        blds16(rx, ok ? xoff[q] + tap : kOob, so, Xs + slot * XR * 64 + (wave + 4 * q) * 512);
                    }
                    // K order: (ky, kx, chunk) as the weight rows are laid out, or chunk-major (chunk, ky, kx) for the layer shapes
                    // the halo-tile 3x3 kernels are bound to (launch_conv sets p.k_chunk_major): one K order per layer
                    if (CHUNK_MAJOR) {
                        if (++kx == p.KW) { kx = 0; if (++ky == p.KH) { ky = 0; ++kc; } }
                    } else if (++kc == kc_n) {
                        kc = 0;
                        if (++kx == p.KW) { kx = 0; ++ky; }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < XP; ++q) blds16(rx2, xoff2[q], (uint32_t)((s - nk1) << 7), Xs + slot * XR * 64 + (wave + 4 * q) * 512);
                }
                {
                    const uint32_t col = CHUNK_MAJOR ? (uint32_t)__builtin_amdgcn_readfirstlane(s < nk1 ? (wtap * p.Cin + (wkc << 6)) * 2 : s << 7)
                                                     : (uint32_t)(s << 7);
#pragma unroll
                    for (int q = 0; q < WP; ++q) blds16(rw, woff[q], col, Ws + slot * BN * 64 + (wave + 4 * q) * 512);
                    if (CHUNK_MAJOR && ++wtap == p.KH * p.KW) { wtap = 0; ++wkc; }
                }
                if (++slot == NSW) { slot = 0; need += 4; }
                if (s >= D) {
                    ring_vmcnt<R::later(0)>();
                    ring_flag_add(fullW + 4 * pslot, lane);
                    if (++pslot == NSW) pslot = 0;
                }
            }
            ring_vmcnt<0>();
            for (int sp = (nsteps > D ? nsteps - D : 0); sp < nsteps; ++sp) {
                ring_flag_add(fullW + 4 * pslot, lane);
                if (++pslot == NSW) pslot = 0;
            }
        }
        // (s_endpgm here, not a return that hipcc merges with the consumers' exit through a flag register: the loader path must
        //  be a leaf of the control-flow graph for tools/isa_check.py to see that no store can reach its counted waits)
        __builtin_amdgcn_endpgm();
    }

    // =========================================== consumer waves ===========================================
    const int cw = wave, wm = cw & 1, wn = cw >> 1;
    const int frow = lane & 15, fq = lane >> 4;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 resv[TM][TN / 2];
    conv_prefetch_residual<TM, TN / 2, WM, WN>(p, resv, m0, n0, wm, wn, frow, fq, M, HoWo);
    // KX3: fragment j is pixel m0 + wm*64 + j*16 + frow; image column 0 kills its kx = 0 tap, column W-1 its kx = 2 tap
    // (kept as bit masks and applied with AND: a select between the loaded fragment and zero on a run-time tap makes hipcc
    //  branch around every ds_read with an `s_waitcnt lgkmcnt(0)` behind each -- cdna_hip_programming.md section 5, trap 4(c))
    uint32_t keep0[TM], keep2[TM]; // all ones, or zero where tap kx = 0 / kx = 2 falls outside the image row
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int x = KX3 ? (m0 + wm * WM + j * 16 + frow) % p.W : 1;
        keep0[j] = (KX3 && x == 0) ? 0u : 0xffffffffu;
        keep2[j] = (KX3 && x == p.W - 1) ? 0u : 0xffffffffu;
    }

    // step s reads weight slot s % NSW (use s / NSW) and activation slot (s / G) % NSX (use (s / G) / NSX), tap kx = s % G
    int wslot = 0, xslot = 0, kx = 0;
    uint32_t wwant = 4, xwant = 4;
    auto wait_step = [&]() { // for the step the cursors point at
        if (RFD_RING_EXP == 2 || RFD_RING_NOSYNC) return;
        if (KX3 ? kx == 0 : false) ring_wait(fullX + 4 * xslot, xwant, dead, gave_up);
        ring_wait(fullW + 4 * wslot, wwant, dead, gave_up);
    };
    auto read_half = [&](int kk, bf16x8 (&A)[TN], bf16x8 (&B)[TM]) {
        if (RFD_RING_EXP == 6 || RFD_RING_EXP == 8) return;
        const bf16_t *ws = Ws + wslot * BN * 64 + (wn * WN) * 64;
        const bf16_t *xs = Xs + xslot * XR * 64 + (wm * WM + (KX3 ? kx : 0)) * 64;
        const int ch = kk * 4 + fq;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int r = i * 16 + frow;
            A[i] = *reinterpret_cast<const bf16x8 *>(ws + r * 64 + ((ch ^ (r & 7)) << 3));
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int r = j * 16 + frow;
            const int rho = r + (KX3 ? kx : 0); // LDS row within the wave's slice (wm * 64 is a multiple of 8): sets the swizzle
            B[j] = *reinterpret_cast<const bf16x8 *>(xs + r * 64 + ((ch ^ (rho & 7)) << 3));
        }
    };
    auto release_and_advance = [&]() { // all fragment reads of the step have been issued: LDS runs them before the adds
        if (!KX3 || kx == G - 1) {
            if (KX3) ring_flag_add(freeX + 4 * xslot, lane);
            if (++xslot == NSX) { xslot = 0; xwant += 4; }
        }
        ring_flag_add(freeW + 4 * wslot, lane);
        if (++wslot == NSW) { wslot = 0; wwant += 4; }
        if (KX3 && ++kx == G) kx = 0;
    };
    auto mfma_half = [&](int s, int kxs, int kk, bf16x8 (&A)[TN], bf16x8 (&B)[TM]) {
        if (KX3) { // taps that fall outside the image row: applied at the point of use, not behind the read (no wait is pulled forward)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const uint32_t k = (kxs == 0 ? keep0[j] : 0xffffffffu) & (kxs == 2 ? keep2[j] : 0xffffffffu); // scalar selects
                u32x4 w = __builtin_bit_cast(u32x4, B[j]);
                w &= k;
                B[j] = __builtin_bit_cast(bf16x8, w);
            }
        }
        if (!KX3 && p.in_scale) {
            // this lane's 8 operand elements are input channels s*64 + kk*32 + fq*8 .. +7 of one pixel
            const float *sc = Sc + s * 64 + kk * 32 + fq * 8;
            const float4 s0 = *reinterpret_cast<const float4 *>(sc), s1 = *reinterpret_cast<const float4 *>(sc + 4);
            const float4 t0 = *reinterpret_cast<const float4 *>(sc + K1), t1 = *reinterpret_cast<const float4 *>(sc + K1 + 4);
            const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float tt[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                bf16x8 v = B[j];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)fmaxf(__builtin_fmaf((float)v[e], ss[e], tt[e]), 0.f); // as conv_igemm_kernel
                B[j] = v;
            }
        }
        if (RFD_RING_EXP == 7 || RFD_RING_EXP == 8) {
#pragma unroll
            for (int i = 0; i < TN; ++i) asm volatile("" : : "v"(A[i]), "v"(B[i])); // keep the reads
            return;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
    };

    // software pipeline over half steps (32 of the 64 K columns): the reads of the next half are issued before the MFMAs of
    // the current one, into the register set the previous half has just left
#if RFD_RING_EXP >= 6
    bf16x8 A0[TN] = {}, B0[TM] = {}, A1[TN] = {}, B1[TM] = {};
#else
    bf16x8 A0[TN], B0[TM], A1[TN], B1[TM];
#endif
    wait_step();
    read_half(0, A0, B0);
    for (int s = 0; s < nsteps; ++s) {
        const int kxs = kx; // tap of step s (the cursors move on below)
        read_half(1, A1, B1);
        release_and_advance();
        mfma_half(s, kxs, 0, A0, B0);
        // (the read is unconditional -- past the last step it fetches a stale slot that nobody uses: with the reads under the
        //  `if`, hipcc has to assume at the join that none were issued and waits lgkmcnt(0) in front of the MFMAs below,
        //  i.e. for the reads issued just above them)
        if (s + 1 < nsteps) wait_step();
        read_half(0, A0, B0);
        mfma_half(s, kxs, 1, A1, B1);
    }
    if (p.fail && ring_flag_read(gave_up) != 0) *p.fail = 1; // some wave of this workgroup stopped waiting: the tile is not to be trusted
    conv_epilogue<TM, TN, WM, WN>(p, acc, resv, m0, n0, wm, wn, frow, fq, M);
    __builtin_amdgcn_endpgm(); // (as the loader path: keeps the two roles disjoint in the emitted control-flow graph)
}

// shapes the ring kernel accepts (launch_conv asks before routing a layer here)
bool conv_ring_supports(const ConvParams &p, bool *kx3)
{
    if (p.Cout % 128 != 0 || p.Cin % 64 != 0 || p.Cin2 % 64 != 0 || p.w1) return false;
    const bool k3 = p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Cin2 == 0 && !p.in_scale && p.Ho == p.H && p.Wo == p.W &&
                    p.W >= 3; // launch_conv's kx_ok: these layers accumulate chunk-major in every kernel that runs them
    if (kx3) *kx3 = k3;
    if (p.in_scale && (p.Cin2 || (size_t)2 * p.KH * p.KW * p.Cin * sizeof(float) > 16 * 1024)) return false;
    return true;
}

int launch_conv_ring(const ConvParams &p, hipStream_t s)
{
    bool kx3 = false;
    if (!conv_ring_supports(p, &kx3)) { set_error("conv_ring: unsupported layer shape"); return RFD_ERR_INVALID_ARG; }
    const int M = p.B * p.Ho * p.Wo;
    const int grid = ceil_div(M, 128) * (p.Cout / 128);
    // the workgroup owns its CU: one ring per CU is the point (rule 4 for the persistent kernels)
    constexpr size_t lds = 160 * 1024;
    static_assert(RingGeom<true>::kSlotBytes + 128 <= lds && RingGeom<false>::kSlotBytes + 128 + 16 * 1024 <= lds, "LDS budget");
#define RFD_RING_LAUNCH(KX, CM)                                                              \
    do {                                                                                     \
        auto kern = conv_ring_kernel<KX, CM>;                                                \
        static DynLdsOnce once;                                                              \
        if (note_launch("conv_ring_kernel<%s, %s>", KX ? "true" : "false", CM ? "true" : "false")) return RFD_OK; \
        RFD_TRY(once.ensure(reinterpret_cast<const void *>(kern), (int)lds));                \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);                          \
    } while (0)
    if (kx3) RFD_RING_LAUNCH(true, true);
    else if (p.k_chunk_major) RFD_RING_LAUNCH(false, true);
    else RFD_RING_LAUNCH(false, false);
#undef RFD_RING_LAUNCH
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

} // namespace rfd

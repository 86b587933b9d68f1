// kernels_post.hip -- everything after the network, on the GPU (gfx950, wave64):
//   decode + threshold + compact   (reference face_detection.rs:319-408, rcnn/anchors.rs:3-21,
//                                   processing/bbox_transform.rs:27-45)
//   deterministic sort             (reference utils.rs:87-95 stable argsort, :97-124 reorder)
//   greedy NMS + gather + rescale  (reference processing/nms.rs:3-65, face_detection.rs:431-493;
//                                   replaces the never-built rcnn/nms_kernel.cu)
//
// All arithmetic is IEEE f32 in the reference's written operation order; this file is compiled
// with -ffp-contract=off so no mul+add is fused.  These kernels are HBM/latency bound integer and
// f32 work -- no MFMA.
#include "kernels.h"

namespace rfd {

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------

// f32 exp with the SAME value the reference's CPU computes: Rust's f32::exp (face_detection.rs:534-535) is the platform libm's
// expf, and glibc's expf (>= 2.27, sysdeps/ieee754/flt-32/e_expf.c: Szabolcs Nagy's exp2f-table algorithm) is not correctly
// rounded -- (float)exp((double)x), which rounds 1-3 shipped, differs from it on about one input in 10^4 by one ulp (round 3: one
// coordinate of 14 752 in the dense-crowd test).  Restated here operation by operation in f64:
//   z = x * (32 / ln 2);  k = nearest integer (ties to even: add and subtract 1.5 * 2^52);  r = z - k, computed with ONE rounding
//   (fma) as the libm build for x86-64 CPUs with FMA does -- the ifunc variant every MI355X host selects; the generic build rounds
//   z first and differs on ~3 inputs in 10^9 (oracle/ test: tests/test_oracle_cpu.py pins this restatement against the host's
//   expf) --; 2^(k/32) from a 32-entry table of the fraction bits plus k's upper bits in the exponent field; degree-3 polynomial
//   in r; one rounding to f32 at the end.  The table = bits(2^(i/32)) - (i << 47), derived with 60-digit arithmetic and equal to
//   glibc's __exp2f_data.tab.  Special cases as e_expf.c: overflow -> +inf, underflow -> 0, NaN propagates.
__device__ __constant__ unsigned long long kExp2fTab[32] = {0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
__device__ __forceinline__ float exp_cr(float x)
{
    const uint32_t abstop = (__float_as_uint(x) >> 20) & 0x7ff;
    if (abstop >= (0x42b00000u >> 20)) { // |x| >= 88 or NaN
        if (__float_as_uint(x) == 0xff800000u) return 0.0f;
        if (abstop >= (0x7f800000u >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);  // > log(2^128)
        if (x < -0x1.9fe368p6f) return 0.0f;                          // < log(2^-150)
    }
    const double xd = (double)x;
    const double InvLn2N = 0x1.71547652b82fep+0 * 32, Shift = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
    double kd = __builtin_fma(InvLn2N, xd, Shift);
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= Shift;
    const double r = __builtin_fma(InvLn2N, xd, -kd);
    const double sc = __longlong_as_double((long long)(kExp2fTab[ki & 31] + (ki << 47)));
    const double z = __builtin_fma(C0, r, C1), r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    return (float)(y * sc);
}

__device__ __forceinline__ uint32_t score_sort_bits(float s)
{
    // monotone map f32 -> u32 (ascending), then inverted so that ascending key = descending score
    if (s == 0.0f) s = 0.0f; // -0.0 == +0.0 under partial_cmp
    uint32_t u = __float_as_uint(s);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    return ~u;
}

// One thread per anchor row g (g = level offset + (h*W + w)*A + a).
// Rows with score >= conf_thr (face_detection.rs:375) are decoded (bbox_pred :516-549,
// clip_boxes bbox_transform.rs:27-45, landmark_pred :551-570) into rows[b][g][0..15] and their sort
// key appended to keys[b][*].  Decoding only the survivors is equivalent to the reference's
// decode-everything-then-select because each row is a pure function of its own inputs.
template <bool NCHW>
__global__ void __launch_bounds__(256) decode_kernel(DecodeParams p)
{
    const int b = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= p.total_anchors) return;
    int l = 0;
    if (g >= p.level_off[1]) l = 1;
    if (g >= p.level_off[2]) l = 2;
    const int r = g - p.level_off[l];
    const int a = r % kA;
    const int pos = r / kA;
    const int fw = p.fw[l], fh = p.fh[l];
    const int h = pos / fw, w = pos - h * fw;
    const int stride = p.stride[l];
    const size_t hw = (size_t)fh * fw;

    float score;
    if (NCHW) // cls [n][2A][h][w]; fg of anchor a = channel A + a (face_detection.rs:322)
        score = p.cls[l][((size_t)b * 2 * kA + kA + a) * hw + pos];
    else      // fused head tensor [n][h][w][32] = cls 2A | bbox 4A | lmk 10A (kernels_conv.hip heads)
        score = p.cls[l][((size_t)b * hw + pos) * 32 + kA + a];
    if (!(score >= p.conf_thr)) return;

    float d[4], ld[10];
    if (NCHW) {
        const float *bb = p.bbox[l] + ((size_t)b * 4 * kA + 4 * a) * hw + pos;
        const float *lm = p.lmk[l] + ((size_t)b * 10 * kA + 10 * a) * hw + pos;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = bb[c * hw];
#pragma unroll
        for (int c = 0; c < 10; ++c) ld[c] = lm[c * hw];
    } else {
        const float *bb = p.cls[l] + ((size_t)b * hw + pos) * 32 + 2 * kA + 4 * a;
        const float *lm = p.cls[l] + ((size_t)b * hw + pos) * 32 + 6 * kA + 10 * a;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = bb[c];
#pragma unroll
        for (int c = 0; c < 10; ++c) ld[c] = lm[c];
    }
    // anchor (rcnn/anchors.rs:9-16): base + (w*stride, h*stride, w*stride, h*stride)
    const float sw = (float)(w * stride), sh = (float)(h * stride);
    const float ax1 = p.base_anchor[l][a][0] + sw, ay1 = p.base_anchor[l][a][1] + sh;
    const float ax2 = p.base_anchor[l][a][2] + sw, ay2 = p.base_anchor[l][a][3] + sh;
    // bbox_pred (face_detection.rs:522-542); bbox_stds = 1 (:366-371)
    const float bw = ax2 - ax1 + 1.0f;
    const float bh = ay2 - ay1 + 1.0f;
    const float cx = ax1 + 0.5f * (bw - 1.0f);
    const float cy = ay1 + 0.5f * (bh - 1.0f);
    const float pcx = d[0] * 1.0f * bw + cx;
    const float pcy = d[1] * 1.0f * bh + cy;
    const float pw = exp_cr(d[2] * 1.0f) * bw;
    const float ph = exp_cr(d[3] * 1.0f) * bh;
    float x1 = pcx - 0.5f * (pw - 1.0f);
    float y1 = pcy - 0.5f * (ph - 1.0f);
    float x2 = pcx + 0.5f * (pw - 1.0f);
    float y2 = pcy + 0.5f * (ph - 1.0f);
    // clip_boxes (bbox_transform.rs:27-45): v.min(hi).max(0); fminf/fmaxf return the non-NaN operand
    const float wmax = (float)p.net_w - 1.0f, hmax = (float)p.net_h - 1.0f;
    x1 = fmaxf(fminf(x1, wmax), 0.0f);
    y1 = fmaxf(fminf(y1, hmax), 0.0f);
    x2 = fmaxf(fminf(x2, wmax), 0.0f);
    y2 = fmaxf(fminf(y2, hmax), 0.0f);

    float4 *row = reinterpret_cast<float4 *>(p.rows + ((size_t)b * p.total_anchors + g) * kDetRow);
    // landmark_pred (face_detection.rs:557-566): from the ANCHOR, not the decoded box; landmark_std = 1
    float lo[10];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        lo[2 * k + 0] = ld[2 * k + 0] * 1.0f * bw + cx;
        lo[2 * k + 1] = ld[2 * k + 1] * 1.0f * bh + cy;
    }
    row[0] = make_float4(x1, y1, x2, y2);
    row[1] = make_float4(score, lo[0], lo[1], lo[2]);
    row[2] = make_float4(lo[3], lo[4], lo[5], lo[6]);
    row[3] = make_float4(lo[7], lo[8], lo[9], 0.0f);

    const int slot = atomicAdd(p.count + b, 1);
    p.keys[(size_t)b * p.total_anchors + slot] = ((uint64_t)score_sort_bits(score) << 32) | (uint32_t)g;
}

int launch_decode(const DecodeParams &p, int n, bool nchw, hipStream_t s)
{
    dim3 grid(ceil_div(p.total_anchors, 256), n);
    if (nchw)
        hipLaunchKernelGGL(decode_kernel<true>, grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(decode_kernel<false>, grid, dim3(256), 0, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

// ------------------------------------------------------------------------------------------------
// sort: the reference stable-sorts the level-concatenated candidates by score descending
// (utils.rs:87-95); level concatenation order == ascending g, so the order is the total order
// (score desc, g asc) = ascending 64-bit key.  Keys are unique, so rank = #{keys smaller} is a
// permutation: no stability argument is needed and the result is deterministic.
//
// Round 3: two kernels instead of the O(n^2) all-pairs rank sort (0.68 ms for 64 dense-crowd images, 1.28 ms at 16 800).
//   chunk_sort_kernel  one 1024-thread workgroup per 1024 keys: bitonic network on one key per thread -- partner exchange by
//                      wave shuffles for distances < 64, through LDS (64-bit reads) for 64..512 -- over the next power of two
//                      >= the chunk's key count.  An image with <= 1024 candidates (every image of the headline workload)
//                      is finished here: the sorted keys and their gathered boxes are written straight out.
//   merge_rank_kernel  dense crowds only: a workgroup takes one sorted chunk (a key per thread), stages the image's other
//                      sorted chunks in LDS (up to 18 x 8 KiB at a time) and adds, per chunk, the branch-free lower bound
//                      of its key -- rank = own index + sum of lower bounds: n log(n) comparisons instead of n^2.
// LDS operands are read as 64-bit words under wave-uniform control flow only (rules (i)-(iii)).
// Both kernels also gather the box of each sorted row into sorted_boxes for the NMS kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kSortChunk = 1024;
constexpr int kSortGroup = 18; // sorted chunks resident in LDS per pass of merge_rank_kernel (144 KiB)

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
    return ((uint64_t)hi << 32) | lo;
}

__global__ void __launch_bounds__(kSortChunk) chunk_sort_kernel(uint64_t *__restrict__ keys, const int *__restrict__ count,
                                                                 const float *__restrict__ rows, uint64_t *__restrict__ sorted_keys,
                                                                 float4 *__restrict__ sorted_boxes, int total_anchors)
{
    __shared__ uint64_t xch[kSortChunk];
    const int b = blockIdx.y, n = count[b], c0 = blockIdx.x * kSortChunk;
    if (c0 >= n) return; // block-uniform
    const int m = min(kSortChunk, n - c0), i = threadIdx.x;
    int P = 64;
    while (P < m) P <<= 1; // block-uniform sort size; slots >= m hold ~0 (larger than every real key) and end up behind them
    uint64_t *k = keys + (size_t)b * total_anchors + c0;
    uint64_t v = i < m ? k[i] : ~0ull;
    for (int kk = 2; kk <= P; kk <<= 1)
        for (int j = kk >> 1; j >= 1; j >>= 1) {
            uint64_t o;
            if (j >= 64) { // block-uniform branch
                xch[i] = v;
                __syncthreads();
                o = xch[i ^ j];
                __syncthreads();
            } else {
                o = shfl_xor_u64(v, j);
            }
            const bool keep_min = ((i & kk) == 0) == ((i & j) == 0);
            v = keep_min ? (o < v ? o : v) : (o > v ? o : v);
        }
    if (i >= m) return;
    if (n <= kSortChunk) { // the whole image: final position = i
        sorted_keys[(size_t)b * total_anchors + i] = v;
        sorted_boxes[(size_t)b * total_anchors + i] =
            *reinterpret_cast<const float4 *>(rows + ((size_t)b * total_anchors + (uint32_t)v) * kDetRow);
    } else {
        k[i] = v; // sorted chunk, in place
    }
}

__global__ void __launch_bounds__(kSortChunk) merge_rank_kernel(const uint64_t *__restrict__ keys, const int *__restrict__ count,
                                                                 const float *__restrict__ rows, uint64_t *__restrict__ sorted_keys,
                                                                 float4 *__restrict__ sorted_boxes, int total_anchors)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_smem[];
    uint64_t *buf = reinterpret_cast<uint64_t *>(sort_smem); // [kSortGroup][kSortChunk]
    const int b = blockIdx.y, n = count[b], c0 = blockIdx.x * kSortChunk;
    if (n <= kSortChunk || c0 >= n) return; // block-uniform: small images were finished by chunk_sort_kernel
    const int nch = (n + kSortChunk - 1) / kSortChunk, i = threadIdx.x;
    const uint64_t *k = keys + (size_t)b * total_anchors;
    const uint64_t mine = c0 + i < n ? k[c0 + i] : ~0ull;
    int rank = i;
    for (int g0 = 0; g0 < nch; g0 += kSortGroup) {
        const int cnt = min(kSortGroup, nch - g0);
        for (int t = i; t < cnt * kSortChunk; t += kSortChunk) {
            const int idx = g0 * kSortChunk + t;
            buf[t] = idx < n ? k[idx] : ~0ull;
        }
        __syncthreads();
        // branch-free lower bound of `mine` in each resident chunk (1024 sorted keys, padding ~0 at the end), two chunks at a
        // time so that the dependent LDS reads of one search overlap the other's
        for (int cc = 0; cc < cnt; cc += 2) {
            const bool two = cc + 1 < cnt;
            const uint64_t *a0 = buf + cc * kSortChunk, *a1 = buf + (two ? cc + 1 : cc) * kSortChunk;
            int p0 = 0, p1 = 0;
#pragma unroll
            for (int step = kSortChunk / 2; step >= 1; step >>= 1) {
                const uint64_t u0 = a0[p0 + step - 1], u1 = a1[p1 + step - 1];
                p0 += u0 < mine ? step : 0;
                p1 += u1 < mine ? step : 0;
            }
            p0 += a0[p0] < mine ? 1 : 0;
            p1 += a1[p1] < mine ? 1 : 0;
            if (g0 + cc != (int)blockIdx.x) rank += p0;                // wave-uniform conditions: the own chunk is skipped
            if (two && g0 + cc + 1 != (int)blockIdx.x) rank += p1;
        }
        __syncthreads();
    }
    if (c0 + i < n) {
        sorted_keys[(size_t)b * total_anchors + rank] = mine;
        sorted_boxes[(size_t)b * total_anchors + rank] =
            *reinterpret_cast<const float4 *>(rows + ((size_t)b * total_anchors + (uint32_t)mine) * kDetRow);
    }
}

int launch_sort(uint64_t *keys, const int *count, const float *rows, uint64_t *sorted_keys,
                float4 *sorted_boxes, int total_anchors, int n, hipStream_t s)
{
    dim3 grid(ceil_div(total_anchors, kSortChunk), n);
    hipLaunchKernelGGL(chunk_sort_kernel, grid, dim3(kSortChunk), 0, s, keys, count, rows, sorted_keys, sorted_boxes, total_anchors);
    RFD_HIP(hipGetLastError());
    if (total_anchors > kSortChunk) { // an image can have more than one chunk of candidates
        const size_t lds = (size_t)kSortGroup * kSortChunk * sizeof(uint64_t);
        static DynLdsOnce once;
        RFD_TRY(once.ensure(reinterpret_cast<const void *>(merge_rank_kernel), (int)lds));
        hipLaunchKernelGGL(merge_rank_kernel, grid, dim3(kSortChunk), lds, s, keys, count, rows, sorted_keys, sorted_boxes, total_anchors);
        RFD_HIP(hipGetLastError());
    }
    return RFD_OK;
}

// ------------------------------------------------------------------------------------------------
// NMS: wave64 bitmask in LDS.
//
// One workgroup per image walks the score-sorted candidates in tiles of 64 (= one wavefront, one
// 64-bit word of the `removed` bitmap held in LDS).  Per tile: wave 0 builds the 64x64 intra-tile
// suppression masks (lane i owns row i), resolves the greedy order serially with wave-uniform bit
// tricks, and emits the kept rows; then all waves test every still-alive later candidate against
// the KEPT boxes of this tile only and publish one ballot word each -- so the work is O(N*K) like
// the reference's CPU loop (nms.rs:10-62), not the O(N^2) all-pairs mask of nms_kernel.cu:34-78,
// and the N x N/64 mask never exists in memory.
//
// Survivor rule: a later box survives iff `ovr <= thresh` (nms.rs:58) -- a NaN overlap suppresses.
// IoU: +1 pixel convention, inter / (area_i + area_j - inter) in f32 (nms.rs:39-54).
// ------------------------------------------------------------------------------------------------
constexpr int kNmsThreads = 1024;
constexpr int kNmsWaves = kNmsThreads / 64;
constexpr int kNmsRegWords = 17;    // bitmap words (64 candidates each) a wave can own in registers
constexpr int kNmsRegCap = kNmsRegWords * kNmsWaves * 64; // 17408 >= the 16800 anchors of a 640x640 input
constexpr int kNmsLdsBoxes = 4096;  // REG = false only: sorted boxes cached in LDS, the rest stream from L2

// the float4 held by lane `src` (wave-uniform) of this wave
__device__ __forceinline__ float4 lane_box(const float4 v, int src)
{
    return make_float4(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.x), src)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.y), src)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.z), src)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.w), src)));
}

// broadcast read of one box from LDS as two 64-bit reads (not one ds_read_b128: see rank_sort_kernel)
__device__ __forceinline__ float4 lds_box_2x64(uint32_t lds_byte_addr)
{
    float2 a, b;
    asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(lds_byte_addr) : "memory");
    return make_float4(a.x, a.y, b.x, b.y);
}

__device__ __forceinline__ float box_area(const float4 b)
{
    return (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
}

__device__ __forceinline__ bool suppresses(const float4 bi, const float area_i, const float4 bj,
                                           const float area_j, const float thresh)
{
    const float xx1 = fmaxf(bi.x, bj.x);
    const float yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z);
    const float yy2 = fminf(bi.w, bj.w);
    float w = xx2 - xx1 + 1.0f;
    float h = yy2 - yy1 + 1.0f;
    w = fmaxf(0.0f, w);
    h = fmaxf(0.0f, h);
    const float inter = w * h;
    const float uni = area_i + area_j - inter;
    const float ovr = inter / uni;
    return !(ovr <= thresh);
}

// REG: every wave keeps the boxes of the bitmap words it owns (w = wave, wave+16, ...) in registers, so
// the suppression scan reads only the 64 kept-candidate boxes of the current tile from LDS.  Used when
// the candidate capacity fits (<= 17408); otherwise boxes are re-read from an LDS cache / L2.
template <bool REG>
__global__ void __launch_bounds__(kNmsThreads) nms_kernel(NmsParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // carve (all offsets multiples of 16)
    float4 *tile_boxes = reinterpret_cast<float4 *>(smem);                                   // [64]
    uint64_t *removed = reinterpret_cast<uint64_t *>(tile_boxes + 64);                       // [nwords_cap]
    uint64_t *keptw = removed + p.nwords_cap;                                                // [nwords_cap]
    int *obase = reinterpret_cast<int *>(keptw + p.nwords_cap);                              // [nwords_cap]
    uint64_t *kept_word = reinterpret_cast<uint64_t *>(obase + p.nwords_cap);                // [2]
    float4 *lds_boxes = reinterpret_cast<float4 *>(kept_word + 2);                           // [kNmsLdsBoxes] (!REG)

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = p.presorted_n >= 0 ? p.presorted_n : p.count[b];
    const int ntiles = (n + 63) >> 6;
    const float4 *sb = p.sorted_boxes + (size_t)b * p.total_anchors;
    const float thresh = p.iou_thr;

    for (int w = tid; w < ntiles; w += kNmsThreads) {
        const int rem = n - w * 64;
        removed[w] = rem >= 64 ? 0ull : ~((1ull << rem) - 1ull); // slots >= n are dead
        keptw[w] = 0ull;
    }
    float4 mybox[REG ? kNmsRegWords : 1];
    if (REG) {
#pragma unroll
        for (int k = 0; k < kNmsRegWords; ++k) {
            const int j = (wave + k * kNmsWaves) * 64 + lane;
            mybox[k] = j < n ? sb[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int j = tid; j < min(n, kNmsLdsBoxes); j += kNmsThreads) lds_boxes[j] = sb[j];
    }
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        // ---- resolve tile t: by the wave that holds its boxes in registers (REG) or by wave 0 ----
        const int owner = REG ? (t & (kNmsWaves - 1)) : 0;
        if (wave == owner) {
            float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
            if (REG) {
                const int kk = t / kNmsWaves; // wave-uniform
#pragma unroll
                for (int k = 0; k < kNmsRegWords; ++k)
                    if (k == kk) box = mybox[k];
            } else {
                const int j = t * 64 + lane;
                if (j < n) box = j < kNmsLdsBoxes ? lds_boxes[j] : sb[j];
            }
            const float area = box_area(box);
            tile_boxes[lane] = box;
            __builtin_amdgcn_wave_barrier();
            const uint64_t rem_t = removed[t];
            const uint64_t alive0 =
                ~(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rem_t >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rem_t));
            uint64_t kept = 0;
            if (alive0 != 0ull) {
                // intra-tile masks: lane i marks every later, still-alive lane jj it would suppress
                uint64_t mask = 0;
                uint64_t cand = alive0;
                while (cand != 0ull) { // wave-uniform walk over the alive candidates only
                    const int jj = __builtin_ctzll(cand);
                    cand &= cand - 1ull;
                    // box jj lives in lane jj of this wave: cross-lane read, no LDS (wide broadcast LDS reads are kept
                    // out of kernels that may share a CU with MFMA conv kernels of another stream, DESIGN.md "concurrency")
                    const float4 other = lane_box(box, jj);
                    if (jj > lane && suppresses(box, area, other, box_area(other), thresh)) mask |= 1ull << jj;
                }
                // greedy resolve in score order; `alive` is wave-uniform
                uint64_t alive = alive0;
                while (alive != 0ull) {
                    const int i = __builtin_ctzll(alive);
                    kept |= 1ull << i;
                    const uint32_t mlo = __builtin_amdgcn_readlane((int)(uint32_t)mask, i);
                    const uint32_t mhi = __builtin_amdgcn_readlane((int)(uint32_t)(mask >> 32), i);
                    const uint64_t mi = ((uint64_t)mhi << 32) | mlo;
                    alive &= ~(mi | (1ull << i));
                }
            }
            if (lane == 0) { kept_word[0] = kept; keptw[t] = kept; }
        }
        __syncthreads();
        // ---- all waves: test every still-alive later candidate against the KEPT boxes of tile t ----
        const uint64_t kmask = kept_word[0];
        if (kmask != 0ull) {
            // every wave owns whole bitmap words: one ballot, one plain LDS store, no atomics
            const uint32_t tile_base = (uint32_t)(uintptr_t)tile_boxes; // LDS byte offset of the tile
            auto scan_word = [&](int w, const float4 bj) {
                const uint64_t rw = removed[w];
                if (rw == ~0ull) return; // wave-uniform
                const float area_j = box_area(bj);
                bool sup = false;
                uint64_t km = kmask;
                while (km != 0ull) { // wave-uniform trip count; ONE LDS broadcast per step (area recomputed:
                    const int i = __builtin_ctzll(km); // a second LDS read per step measured 45 % slower)
                    km &= km - 1ull;
                    const float4 bi = lds_box_2x64(tile_base + (uint32_t)i * 16u);
                    sup |= suppresses(bi, box_area(bi), bj, area_j, thresh);
                }
                const uint64_t bal = __ballot(sup);
                if (lane == 0 && (bal & ~rw) != 0ull) removed[w] = rw | bal;
            };
            if (REG) {
#pragma unroll
                for (int k = 0; k < kNmsRegWords; ++k) {
                    const int w = wave + k * kNmsWaves;
                    if (w > t && w < ntiles) scan_word(w, mybox[k]);
                }
            } else {
                for (int w = t + 1 + wave; w < ntiles; w += kNmsWaves) {
                    const int j = w * 64 + lane;
                    float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j < n) bj = j < kNmsLdsBoxes ? lds_boxes[j] : sb[j];
                    scan_word(w, bj);
                }
            }
        }
        __syncthreads();
    }

    // ---- emit the kept rows in score order (face_detection.rs:433-464), rescaled (:473-493): exclusive
    //      prefix of the per-word kept counts (wave 0), then every thread gathers its own candidates ----
    if (wave == 0) {
        const int per = (ntiles + 63) >> 6;
        int sum = 0;
        for (int k = 0; k < per; ++k) {
            const int w = lane * per + k;
            if (w < ntiles) sum += __builtin_popcountll(keptw[w]);
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        int run = incl - sum;
        for (int k = 0; k < per; ++k) {
            const int w = lane * per + k;
            if (w < ntiles) { obase[w] = run; run += __builtin_popcountll(keptw[w]); }
        }
        if (lane == 63) {
            if (p.out_total) p.out_total[b] = incl;
            if (p.out_count) p.out_count[b] = min(incl, p.max_det);
        }
    }
    __syncthreads();
    for (int j = tid; j < n; j += kNmsThreads) {
        const uint64_t kw = keptw[j >> 6];
        if (!((kw >> (j & 63)) & 1ull)) continue;
        const int o = obase[j >> 6] + __builtin_popcountll(kw & ((1ull << (j & 63)) - 1ull));
        if (o >= p.max_det) continue;
        uint32_t g = (uint32_t)j;
        if (p.rows) {
            g = (uint32_t)p.sorted_keys[(size_t)b * p.total_anchors + j];
            float *ob = p.out_boxes + ((size_t)b * p.max_det + o) * 5;
            const float4 *row = reinterpret_cast<const float4 *>(p.rows + ((size_t)b * p.total_anchors + g) * kDetRow);
            const float sc = p.det_scale[b];
            const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
            ob[0] = r0.x / sc; ob[1] = r0.y / sc; ob[2] = r0.z / sc; ob[3] = r0.w / sc;
            ob[4] = r1.x;
            float *ol = p.out_lmk + ((size_t)b * p.max_det + o) * 10;
            ol[0] = r1.y / sc; ol[1] = r1.z / sc; ol[2] = r1.w / sc;
            ol[3] = r2.x / sc; ol[4] = r2.y / sc; ol[5] = r2.z / sc; ol[6] = r2.w / sc;
            ol[7] = r3.x / sc; ol[8] = r3.y / sc; ol[9] = r3.z / sc;
        }
        if (p.out_gidx) p.out_gidx[(size_t)b * p.max_det + o] = (int)g;
    }
}

// ------------------------------------------------------------------------------------------------
// Dense crowds (BASELINE.json configs[4]: ~12 k candidates, ~2 k faces per image): one workgroup per image is VALU-bound
// (every later candidate against every kept box: 64 of 256 CUs busy for 4 ms at B = 64).  Greedy NMS is sequential in score
// order, but only through the KEPT boxes: the sorted list is cut into kNmsChunks runs of tiles, one workgroup each.
// Chunk c
//   A. waits for chunks 0 .. c-1 one after the other (a flag per chunk in global memory, release / acquire at device scope)
//      and tests its own candidates against their published kept boxes -- independent work, it overlaps the predecessors'
//      own NMS except for the last of them;
//   B. runs the tile loop of nms_kernel on what is left of its own run of tiles;
//   C. appends its kept boxes to the image's list, publishes the count, emits its rows behind the earlier chunks' rows.
// The critical path is the sum of the chunks' INTERNAL loops, each over a quarter of the tiles with a quarter of the words
// to scan per tile.  The result is the same greedy sequence: a candidate is tested against exactly the kept boxes that
// precede it in score order, with the same f32 arithmetic.
// Deadlock: a workgroup waits only for workgroups that drew a LOWER ticket in the same launch, i.e. that are already running
// (the chunk id is the ticket, not blockIdx); the spin is bounded anyway (spin_fail).  Images with few candidates (<= kNmsChunkMin) are done by chunk 0 alone.
// ------------------------------------------------------------------------------------------------
constexpr int kNmsChunkMin = 2048;
// bitmap words a wave owns in registers: a chunk has at most ceil(272 / kNmsChunks) = 68 tiles (17408 candidates), an unsplit
// image kNmsChunkMin / 64 = 32
constexpr int kNmsChunkWords = (((kNmsRegCap / 64 + kNmsChunks - 1) / kNmsChunks) + kNmsWaves - 1) / kNmsWaves;
static_assert(kNmsChunkMin / 64 <= kNmsChunkWords * kNmsWaves, "unsplit images must fit the chunk kernel's registers");
__global__ void __launch_bounds__(kNmsThreads) nms_chunked_kernel(NmsParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile_boxes = reinterpret_cast<float4 *>(smem);                    // [64]
    uint64_t *removed = reinterpret_cast<uint64_t *>(tile_boxes + 64);        // [nwords_cap] (local word index)
    uint64_t *keptw = removed + p.nwords_cap;                                 // [nwords_cap]
    int *obase = reinterpret_cast<int *>(keptw + p.nwords_cap);               // [nwords_cap]
    uint64_t *kept_word = reinterpret_cast<uint64_t *>(obase + p.nwords_cap); // [2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Which (image, chunk) this workgroup is comes from a TICKET drawn when it starts to run, not from blockIdx: a chunk waits for
    // the chunks of its image with lower tickets, and a lower ticket has been drawn, so its workgroup is running -- forward
    // progress by construction, whatever order the dispatcher starts workgroups in (round-3 review; rounds 2-3 relied on
    // "lower blockIdx starts first", which HIP does not promise).  ONE returning atomic add per workgroup (a compare-and-swap
    // loop on an {epoch, count} word serialised the 128 workgroups of a 32-image launch: 0.12 ms instead of 0.02).  Two counters
    // alternate between launches (p.ticket_sel, toggled by the host): the workgroup that draws ticket 0 zeroes the OTHER one for
    // the next launch -- every launch that used it has completed, launches of a context being ordered on its stream.
    if (tid == 0) {
        const unsigned mine = __hip_atomic_fetch_add(p.ticket + p.ticket_sel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mine == 0u) __hip_atomic_store(p.ticket + (p.ticket_sel ^ 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reinterpret_cast<int *>(kept_word)[0] = (int)mine;
    }
    __syncthreads();
    const int vb = __builtin_amdgcn_readfirstlane(reinterpret_cast<int *>(kept_word)[0]);
    __syncthreads(); // kept_word is reused below
    const int b = vb / kNmsChunks, c = vb % kNmsChunks;
    const int n = p.count[b];
    const int ntiles = (n + 63) >> 6;
    const bool split = n > kNmsChunkMin;
    if (!split && c > 0) return;
    const int tpc = split ? (ntiles + kNmsChunks - 1) / kNmsChunks : ntiles;
    const int t0 = c * tpc, t1 = min(ntiles, t0 + tpc), nloc = t1 - t0;
    if (split && nloc <= 0) return; // empty tail chunk: nobody waits for it
    const float4 *sb = p.sorted_boxes + (size_t)b * p.total_anchors;
    float4 *kb = p.kept_boxes + (size_t)b * p.total_anchors;
    const float thresh = p.iou_thr;

    for (int lw = tid; lw < nloc; lw += kNmsThreads) {
        const int rem = n - (t0 + lw) * 64;
        removed[lw] = rem >= 64 ? 0ull : ~((1ull << rem) - 1ull); // slots >= n are dead
        keptw[lw] = 0ull;
    }
    float4 mybox[kNmsChunkWords]; // local word lw = wave + 16 k
#pragma unroll
    for (int k = 0; k < kNmsChunkWords; ++k) {
        const int j = (t0 + wave + k * kNmsWaves) * 64 + lane;
        mybox[k] = (wave + k * kNmsWaves < nloc && j < n) ? sb[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    const uint32_t tile_base = (uint32_t)(uintptr_t)tile_boxes;
    float myarea[kNmsChunkWords];
#pragma unroll
    for (int k = 0; k < kNmsChunkWords; ++k) myarea[k] = box_area(mybox[k]);
    // Test this wave's words lw > lw_min against the kept boxes `kmask` of tile_boxes.  Kept box OUTER, the (at most five)
    // words inner: one LDS broadcast per kept box instead of one per (kept box, word) pair, the words' boxes and areas in
    // registers -- the loop is VALU-bound, not LDS-latency-bound.  Every wave owns whole bitmap words: one ballot, one plain LDS
    // store per word, no atomics (as nms_kernel).
    auto scan_words = [&](int lw_min, uint64_t kmask) {
        uint32_t act = 0; // wave-uniform: words still to be scanned
#pragma unroll
        for (int k = 0; k < kNmsChunkWords; ++k) {
            const int lw = wave + k * kNmsWaves;
            if (lw > lw_min && lw < nloc) {
                const uint64_t rw = removed[lw];
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rw);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rw >> 32));
                if ((lo & hi) != 0xffffffffu) act |= 1u << k;
            }
        }
        if (act == 0u) return;
        uint32_t supb = 0; // bit k: this lane's candidate of word k is suppressed
        uint64_t km = kmask;
        while (km != 0ull) {
            const int i = __builtin_ctzll(km);
            km &= km - 1ull;
            const float4 bi = lds_box_2x64(tile_base + (uint32_t)i * 16u);
            const float area_i = box_area(bi);
#pragma unroll
            for (int k = 0; k < kNmsChunkWords; ++k)
                if (act & (1u << k)) supb |= (uint32_t)suppresses(bi, area_i, mybox[k], myarea[k], thresh) << k;
        }
#pragma unroll
        for (int k = 0; k < kNmsChunkWords; ++k)
            if (act & (1u << k)) {
                const int lw = wave + k * kNmsWaves;
                const uint64_t rw = removed[lw];
                const uint64_t bal = __ballot((supb >> k) & 1u);
                if (lane == 0 && (bal & ~rw) != 0ull) removed[lw] = rw | bal;
            }
    };

    // ---- A: the kept boxes of the earlier chunks, in order, AS THEY ARE PUBLISHED (a chunk appends the kept boxes of every
    //      tile it resolves and advances its progress word {epoch : 32 | done : 1 | count : 31}) ----
    unsigned long long *prog = reinterpret_cast<unsigned long long *>(p.chunk_state) + (size_t)b * kNmsChunks;
    int *sh = reinterpret_cast<int *>(kept_word); // [0] kept boxes available, [1] predecessor done (between barriers)
    int off = 0;
    for (int e = 0; e < c; ++e) {
        int consumed = 0;
        while (true) {
            if (tid == 0) {
                int it = 0, avail = 0, done = 0;
                while (true) {
                    const unsigned long long v = __hip_atomic_load(&prog[e], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(v >> 32) == p.epoch) {
                        avail = (int)(v & 0x7fffffffull);
                        done = (int)((v >> 31) & 1ull);
                    }
                    if (avail > consumed || done) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++it > (1 << 22)) { *p.spin_fail = 1; done = 1; break; }
                }
                sh[0] = avail;
                sh[1] = done;
            }
            __syncthreads();
            (void)__hip_atomic_load(&prog[e], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); // every wave: see the boxes
            const int avail = sh[0], done = sh[1];
            while (consumed < avail) {
                const int nb = min(64, avail - consumed);
                if (tid < 64) tile_boxes[tid] = tid < nb ? kb[off + consumed + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
                __syncthreads();
                scan_words(-1, nb >= 64 ? ~0ull : (1ull << nb) - 1ull);
                __syncthreads();
                consumed += nb;
            }
            __syncthreads(); // sh is rewritten by the next poll
            if (done) break;
        }
        off += consumed;
    }

    // ---- B: the tile loop of nms_kernel<true> over this chunk's tiles (local word index lt) ----
    const bool publish = split && t1 < ntiles; // somebody follows
    int npub = 0;                               // kept boxes of this chunk published so far (wave-uniform, same in every wave)
    for (int lt = 0; lt < nloc; ++lt) {
        const int owner = lt & (kNmsWaves - 1);
        if (wave == owner) {
            float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
            const int kk = lt / kNmsWaves; // wave-uniform
#pragma unroll
            for (int k = 0; k < kNmsChunkWords; ++k)
                if (k == kk) box = mybox[k];
            const float area = box_area(box);
            tile_boxes[lane] = box;
            __builtin_amdgcn_wave_barrier();
            const uint64_t rem_t = removed[lt];
            const uint64_t alive0 =
                ~(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rem_t >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rem_t));
            uint64_t kept = 0;
            if (alive0 != 0ull) {
                uint64_t mask = 0;
                uint64_t cand = alive0;
                while (cand != 0ull) {
                    const int jj = __builtin_ctzll(cand);
                    cand &= cand - 1ull;
                    const float4 other = lane_box(box, jj);
                    if (jj > lane && suppresses(box, area, other, box_area(other), thresh)) mask |= 1ull << jj;
                }
                uint64_t alive = alive0;
                while (alive != 0ull) {
                    const int i = __builtin_ctzll(alive);
                    kept |= 1ull << i;
                    const uint32_t mlo = __builtin_amdgcn_readlane((int)(uint32_t)mask, i);
                    const uint32_t mhi = __builtin_amdgcn_readlane((int)(uint32_t)(mask >> 32), i);
                    const uint64_t mi = ((uint64_t)mhi << 32) | mlo;
                    alive &= ~(mi | (1ull << i));
                }
            }
            if (lane == 0) { kept_word[0] = kept; keptw[lt] = kept; }
            if (publish) { // append this tile's kept boxes to the image's list and let the later chunks see them
                const int nk = __builtin_popcountll(kept);
                if ((kept >> lane) & 1ull) kb[off + npub + __builtin_popcountll(kept & ((1ull << lane) - 1ull))] = box;
                npub += nk;
                if (nk > 0 || lt + 1 == nloc) {
                    __threadfence();
                    if (lane == 0)
                        __hip_atomic_store(&prog[c], ((unsigned long long)(unsigned)p.epoch << 32) |
                                                         (lt + 1 == nloc ? 0x80000000ull : 0ull) | (unsigned long long)npub,
                                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
        const uint64_t kmask = kept_word[0];
        if (publish && wave != owner) npub += __builtin_popcountll(kmask); // every wave tracks the published count
        if (kmask != 0ull) scan_words(lt, kmask);
        __syncthreads();
    }

    // ---- C: exclusive prefix of the per-word kept counts (wave 0), the chunk's kept boxes appended to the image's list ----
    if (wave == 0) {
        const int per = (nloc + 63) >> 6;
        int sum = 0;
        for (int k = 0; k < per; ++k) {
            const int lw = lane * per + k;
            if (lw < nloc) sum += __builtin_popcountll(keptw[lw]);
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        int run = incl - sum;
        for (int k = 0; k < per; ++k) {
            const int lw = lane * per + k;
            if (lw < nloc) { obase[lw] = run; run += __builtin_popcountll(keptw[lw]); }
        }
        if (lane == 63) {
            kept_word[1] = (uint64_t)incl; // the chunk's kept count
            if (t1 == ntiles) {            // the chunk that ends the list knows the image's total
                if (p.out_total) p.out_total[b] = off + incl;
                if (p.out_count) p.out_count[b] = min(off + incl, p.max_det);
            }
        }
    }
    __syncthreads();
    const int jlo = t0 * 64, jhi = min(n, t1 * 64);
    // ---- emit the kept rows in score order behind the earlier chunks' (face_detection.rs:433-464, :473-493) ----
    for (int j = jlo + tid; j < jhi; j += kNmsThreads) {
        const int lw = (j >> 6) - t0;
        const uint64_t kw = keptw[lw];
        if (!((kw >> (j & 63)) & 1ull)) continue;
        const int o = off + obase[lw] + __builtin_popcountll(kw & ((1ull << (j & 63)) - 1ull));
        if (o >= p.max_det) continue;
        uint32_t g = (uint32_t)j;
        if (p.rows) {
            g = (uint32_t)p.sorted_keys[(size_t)b * p.total_anchors + j];
            float *ob = p.out_boxes + ((size_t)b * p.max_det + o) * 5;
            const float4 *row = reinterpret_cast<const float4 *>(p.rows + ((size_t)b * p.total_anchors + g) * kDetRow);
            const float sc = p.det_scale[b];
            const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
            ob[0] = r0.x / sc; ob[1] = r0.y / sc; ob[2] = r0.z / sc; ob[3] = r0.w / sc;
            ob[4] = r1.x;
            float *ol = p.out_lmk + ((size_t)b * p.max_det + o) * 10;
            ol[0] = r1.y / sc; ol[1] = r1.z / sc; ol[2] = r1.w / sc;
            ol[3] = r2.x / sc; ol[4] = r2.y / sc; ol[5] = r2.z / sc; ol[6] = r2.w / sc;
            ol[7] = r3.x / sc; ol[8] = r3.y / sc; ol[9] = r3.z / sc;
        }
        if (p.out_gidx) p.out_gidx[(size_t)b * p.max_det + o] = (int)g;
    }
}

// ------------------------------------------------------------------------------------------------
// FaceSelection::call (reference src/pipeline/module/face_selection.rs:72-189) as
// an optional device epilogue: one thread per image walks that image's kept rows in the reference's loop
// order (strict `>` keeps the first maximum; the key points are those of the first row within 2 px of
// the chosen box), so the result is bit-identical to the CPU logic by construction and only 16 floats
// per image have to leave the GPU.  Scalar work on <= max_det rows: latency-, not bandwidth-bound.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) face_select_kernel(SelectParams p)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= p.n) return;
    const int k = p.count[b];
    const float *boxes = p.boxes + (size_t)b * p.max_det * 5;
    const float *kps = p.lmk + (size_t)b * p.max_det * 10;
    float *ob = p.out_box + (size_t)b * 5, *ok = p.out_kps + (size_t)b * 10;
    int sel = -1, kp_row = -1;
    if (p.is_enroll) { // get_biggest_area_face :28-53
        float biggest = 0.0f;
        for (int i = 0; i < k; ++i) {
            const float *r = boxes + 5 * i;
            const float a = (r[2] - r[0]) * (r[3] - r[1]);
            if (a > biggest) { biggest = a; sel = i; }
        }
        kp_row = sel;
    } else {
        const float W = (float)p.img_w[b], H = (float)p.img_h[b];
        const float mcl = p.margin_center_left_ratio * W, mcr = p.margin_center_right_ratio * W; // :107-108
        const float me = fminf(50.0f, p.margin_edge_ratio * W);                                   // :109-110
        const float x_cen = W / 2.0f;
        int n_valid = 0, n_center = 0;
        for (int i = 0; i < k; ++i) { // :114-139
            const float *d = boxes + 5 * i;
            const float area = (d[2] - d[0]) * (d[2] - d[0]); // sic (:113)
            const float bcw = (d[0] + d[2]) / 2.0f, bch = (d[1] + d[3]) / 2.0f;
            const bool valid = bcw >= me && bcw <= W - me && bch >= me && bch <= H - me && area / (H * W) >= p.minimum_face_ratio;
            n_valid += valid;
            n_center += valid && (-mcl <= bcw - x_cen && bcw - x_cen <= mcr);
        }
        const int mode = n_center ? 2 : (n_valid ? 1 : 0); // pool: centre boxes, else valid boxes, else all (:141-147)
        float max_size = 0.0f;
        for (int i = 0; i < k; ++i) { // :152-158
            const float *d = boxes + 5 * i;
            if (mode) {
                const float area = (d[2] - d[0]) * (d[2] - d[0]);
                const float bcw = (d[0] + d[2]) / 2.0f, bch = (d[1] + d[3]) / 2.0f;
                const bool valid = bcw >= me && bcw <= W - me && bch >= me && bch <= H - me && area / (H * W) >= p.minimum_face_ratio;
                if (!valid) continue;
                if (mode == 2 && !(-mcl <= bcw - x_cen && bcw - x_cen <= mcr)) continue;
            }
            const float tem = (d[2] - d[0]) + (d[3] - d[1]);
            if (tem > max_size) { max_size = tem; sel = i; }
        }
        if (sel >= 0) { // :163-180
            const float *o = boxes + 5 * sel;
            for (int i = 0; i < k; ++i) {
                const float *r = boxes + 5 * i;
                if (fabsf(o[0] - r[0]) <= 2.0f && fabsf(o[1] - r[1]) <= 2.0f && fabsf(o[2] - r[2]) <= 2.0f && fabsf(o[3] - r[3]) <= 2.0f) {
                    kp_row = i;
                    break;
                }
            }
        }
    }
    int found = 0;
    if (sel >= 0) {
        found = 1;
        for (int c = 0; c < 5; ++c) ob[c] = boxes[5 * sel + c];
        if (kp_row >= 0) {
            found = 3;
            for (int c = 0; c < 10; ++c) ok[c] = kps[10 * kp_row + c];
        }
    }
    p.out_found[b] = found;
}

int launch_face_select(const SelectParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(face_select_kernel, dim3(ceil_div(p.n, 64)), dim3(64), 0, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

size_t nms_lds_bytes(int total_anchors, bool reg)
{
    const int nwords = ceil_div(total_anchors, 64);
    const int nwords_cap = (nwords + 3) & ~3;
    return (size_t)64 * sizeof(float4) + (size_t)(2 * nwords_cap + 2) * sizeof(uint64_t) +
           (size_t)nwords_cap * sizeof(int) + (reg ? 0 : (size_t)kNmsLdsBoxes * sizeof(float4));
}

int launch_nms(NmsParams p, int n_images, hipStream_t s, bool *used_chunked)
{
    const int nwords = ceil_div(p.total_anchors, 64);
    p.nwords_cap = (nwords + 3) & ~3;
    const bool reg = p.total_anchors <= kNmsRegCap;
    const size_t lds = nms_lds_bytes(p.total_anchors, reg);
    if (lds > 160 * 1024) {
        set_error("NMS bitmap for %d anchors does not fit LDS", p.total_anchors);
        return RFD_ERR_CAPACITY;
    }
    static DynLdsOnce once_stream, once_reg;
    RFD_TRY(once_stream.ensure(reinterpret_cast<const void *>(nms_kernel<false>), 160 * 1024));
    RFD_TRY(once_reg.ensure(reinterpret_cast<const void *>(nms_kernel<true>), 160 * 1024));
    const bool chunked = reg && p.kept_boxes && p.chunk_state && p.spin_fail && p.ticket && p.epoch > 0 && p.presorted_n < 0;
    if (used_chunked) *used_chunked = chunked; // the caller alternates the ticket counter only when one was drawn from
    if (chunked)
        hipLaunchKernelGGL(nms_chunked_kernel, dim3(n_images * kNmsChunks), dim3(kNmsThreads), lds, s, p);
    else if (reg) hipLaunchKernelGGL(nms_kernel<true>, dim3(n_images), dim3(kNmsThreads), lds, s, p);
    else hipLaunchKernelGGL(nms_kernel<false>, dim3(n_images), dim3(kNmsThreads), lds, s, p);
    RFD_HIP(hipGetLastError());
    return RFD_OK;
}

} // namespace rfd

// rfd_common.h -- shared host/device helpers for librfd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <string>

#include "../../include/rfd.h"

namespace rfd {

// ---- error plumbing: every HIP failure is returned, never printed-and-ignored
//      (the reference's CUDA_CHECK only prints, src/rcnn/nms_kernel.cu:12-19) ----
void set_error(const char *fmt, ...);
const char *get_error();

struct HipError {
    hipError_t code;
    const char *what;
    const char *file;
    int line;
};

#define RFD_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::rfd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                             __LINE__);                                                   \
            return RFD_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)

#define RFD_CHECK_ARG(cond, msg)                                    \
    do {                                                            \
        if (!(cond)) {                                              \
            ::rfd::set_error("invalid argument: %s (%s)", msg, #cond); \
            return RFD_ERR_INVALID_ARG;                             \
        }                                                           \
    } while (0)

#define RFD_TRY(expr)            \
    do {                         \
        int _s = (expr);         \
        if (_s != RFD_OK) return _s; \
    } while (0)

// ---- bf16 as raw 16-bit (device storage type); round-to-nearest-even on the host ----
typedef uint16_t bf16_t;

__host__ static inline bf16_t f32_to_bf16_host(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40); // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ static inline float bf16_to_f32_host(bf16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ---- detection geometry constants (reference face_detection.rs:52-80) ----
constexpr int kNumLevels = 3;
constexpr int kStrides[kNumLevels] = {32, 16, 8}; // _feat_stride_fpn, face_detection.rs:52
constexpr int kA = 2;                             // anchors per position (_num_anchors)
constexpr int kDetRow = 16;                       // floats per decoded row: box4 score1 lmk10 pad1

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- kernel-choice introspection (rfd_debug_op_kernels): with `dry` set on the calling thread, every launch path records the
//      kernel it WOULD launch (the name rocprofv3 reports, without the rfd:: prefix) and returns without launching.  The tools
//      that attribute time / traffic to kernels (tools/traffic_model.py, tools/roof_gap.py) ask the library instead of
//      mirroring launch_conv()'s rules (round-3 review: the mirror had gone stale).
struct LaunchNote {
    bool dry = false;
    std::string names; // " + "-separated
};
LaunchNote &launch_note();
bool note_launch(const char *fmt, ...); // true: dry run, the caller returns RFD_OK without launching

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: one flag per (kernel, device),
// read and set atomically (the attribute call itself is idempotent, so two threads racing through it is harmless).
// Usage: `static DynLdsOnce once; RFD_TRY(once.ensure(kernel_ptr, bytes));` next to the launch.
struct DynLdsOnce {
    static constexpr int kMaxDevices = 64;
    std::atomic<int> bytes_set[kMaxDevices] = {};
    int ensure(const void *kernel, int bytes)
    {
        int dev = 0;
        RFD_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= kMaxDevices) { set_error("device ordinal %d out of range", dev); return RFD_ERR_NO_DEVICE; }
        if (bytes_set[dev].load(std::memory_order_acquire) >= bytes) return RFD_OK;
        RFD_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        bytes_set[dev].store(bytes, std::memory_order_release);
        return RFD_OK;
    }
};

} // namespace rfd

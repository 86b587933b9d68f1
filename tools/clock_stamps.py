#!/usr/bin/env python3
"""The clock the chip holds INSIDE each conv kernel class while the real pass runs (B = 32, two chains, async 2), from a diagnostic
build with one pair of s_memtime / s_memrealtime stamps per workgroup (tools/build_variant.sh clk -DRFD_CLOCK_STAMPS).
MI355X_MICROARCH.md, "DVFS give-back": the chip lowers its clock under MFMA load, so the spec's 2.5 PF/s (at 2.4 GHz) is not what a
kernel's cycles are worth.  Prints, per kernel class, clock = sum(d s_memtime) / sum(d s_memrealtime) x 100 MHz over every workgroup
of the last SECONDS of back-to-back passes, and the img/s of that window.
usage: RFD_HIP_LIB=tools/bin/librfd_hip_clk.so python tools/clock_stamps.py [seconds]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import helpers  # noqa: E402
import rfd_hip  # noqa: E402
from rfd_hip import parallel  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
B = 32
NAMES = ["conv_igemm_kernel", "pw_stream_kernel", "pw_b2b_kernel", "pw_pair_kernel", "conv3x3_kx_kernel", "conv3x3_c64_kernel",
         "conv3x3_halo_kernel", "pw_gemm_kernel", "pw_wide_kernel", "conv_b2b_s1_kernel", "conv_b2b_s1_persistent_kernel",
         "conv_b2b_s1_persistent_k128_kernel", "stem_persistent_kernel"]
dev = torch.device("cuda", 0)
det = rfd_hip.RetinaFaceDetection(image_size=(640, 640), max_batch_size=B, max_det=1024)
det.init_synthetic_weights(1234)
frames = torch.from_numpy(np.stack([helpers.make_image(1000 + i, 640, 640) for i in range(B)])).to(dev)
fptrs = [frames.data_ptr() + i * 640 * 640 * 3 for i in range(B)]
shapes = [(640, 640)] * B
slab = parallel.DetectionSlab(B, 1024, device=dev)
pb, pl, pc, pt = slab.pointers()
det.set_stream(torch.cuda.current_stream().cuda_stream)
det.detect_device(fptrs, shapes, pb, pl, pc, pt, async_=0)  # stream tuner
L = rfd_hip.load_library()
L.rfd_debug_clock_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 32)()


def run(seconds):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            det.detect_device(fptrs, shapes, pb, pl, pc, pt, async_=2)
        det.sync()
        n += 10
    return n * B / (time.perf_counter() - t0)


run(2.0)                                  # >= 2 s of back-to-back launches before the window that counts
det.sync(); torch.cuda.synchronize()
assert L.rfd_debug_clock_stamps(buf, 1) == 0
rate = run(SECONDS)
det.sync(); torch.cuda.synchronize()
assert L.rfd_debug_clock_stamps(buf, 0) == 0
v = np.array(list(buf), np.float64).reshape(16, 2)
print("B = %d, %.1f s window, %.0f img/s (instrumented build)" % (B, SECONDS, rate))
tot = v.sum(0)
for n, (dt, dr) in zip(NAMES, v):
    if dr > 0:
        print("  %-38s %5.2f GHz   (%4.1f %% of the stamped workgroup time)" % (n, dt / dr * 0.1, 100 * dr / tot[1]))
print("  %-38s %5.2f GHz" % ("all stamped workgroups", tot[0] / tot[1] * 0.1))

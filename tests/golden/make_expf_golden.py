#!/usr/bin/env python3
"""Golden vectors for the decode kernel's `exp` (f32::exp at face_detection.rs:534-535 = the platform libm's expf): 8 192 inputs and
the outputs of THIS container's libm (glibc 2.35, x86-64 with FMA: the ifunc variant __expf_fma), through oracle/librfd_oracle.so's
rfd_oracle_expf_libm.  tests/test_oracle_cpu.py checks the device algorithm's CPU twin (rfd_oracle_expf_restated) against them, so
the restatement stays pinned on hosts whose libm differs.  usage: python tests/golden/make_expf_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(20261005)
x = np.concatenate([
    rng.normal(0, 0.6, 4096),                 # the box-delta range (dw, dh of a trained detector)
    rng.uniform(-6, 6, 2048),
    rng.uniform(-104, 89, 1024),              # the whole finite range of expf
    np.array([0.0, -0.0, 88.0, 88.7, 88.72283, 88.72284, 88.8, -87.3, -87.4, -103.9, -103.97, -104.0, 1e-8, -1e-8, 1e-40]),
]).astype(np.float32)
pad = 8192 - x.size
x = np.concatenate([x, rng.uniform(-1, 1, pad).astype(np.float32)])
y = O.expf(x)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "expf_libm_glibc235.npz"), x=x, y=y)
print("wrote", x.size, "vectors; libm:", os.popen("ldd --version | head -1").read().strip())

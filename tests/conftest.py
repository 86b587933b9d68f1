import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def rfd():
    # torch bundles its own libamdhip64; loading it FIRST makes librfd_hip.so bind to that same copy (same SONAME).
    # The other order puts two HIP runtimes into one process, and whichever initialises second may find no device.
    import torch  # noqa: F401
    import rfd_hip
    if not os.path.exists(rfd_hip.LIB_PATH):  # fresh checkout: the .so is git-ignored; hipcc cross-compiles without a GPU
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "rs-face-detection_amd", "build.sh")])
    rfd_hip.load_library()  # raises if librfd_hip.so is missing: no CPU fallback
    return rfd_hip

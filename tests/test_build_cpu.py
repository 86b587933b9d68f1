"""Build-product checks that need no GPU: no device kernel of the library may spill registers to scratch.

Why a test: the LDS-DMA kernels sequence their staging with `s_waitcnt vmcnt`, and a scratch reload is a vector-memory load
on the same counter -- the compiler then drains it (`vmcnt(0)`) in the middle of the pipeline.  One runtime flag too many
in conv_igemm_kernel did exactly that in round 2: 36 scratch instructions, every 1x1 layer 2x slower, results unchanged."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "rs-face-detection_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_notes(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "dev.co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels = []
    for block in txt.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
        kernels.append((name, get("private_segment_fixed_size"), get("vgpr_spill_count"), get("sgpr_spill_count"), get("vgpr_count")))
    return kernels


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang-offload-bundler")), reason="LLVM offload tools not installed")
def test_no_device_kernel_spills_to_scratch(tmp_path):
    objs = [os.path.join(BUILD, f + ".o") for f in ("kernels_pre", "kernels_post", "kernels_conv")]
    missing = [o for o in objs if not os.path.exists(o)]
    assert not missing, "run rs-face-detection_amd/build.sh (or __graft_entry__.build()) first: %s" % missing
    total = 0
    for o in objs:
        for name, scratch, vsp, ssp, vgpr in _kernel_notes(o, str(tmp_path)):
            total += 1
            assert scratch == 0 and vsp == 0, "%s: %d bytes of scratch, %d VGPR spills (vgpr_count %d)" % (name, scratch, vsp, vgpr)
    assert total >= 40   # pre/post kernels + every conv instantiation

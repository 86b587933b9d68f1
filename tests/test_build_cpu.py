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
    objs = [os.path.join(BUILD, f + ".o") for f in ("kernels_pre", "kernels_post", "kernels_conv", "kernels_ring", "kernels_f32")]
    missing = [o for o in objs if not os.path.exists(o)]
    assert not missing, "run rs-face-detection_amd/build.sh (or __graft_entry__.build()) first: %s" % missing
    total = 0
    for o in objs:
        for name, scratch, vsp, ssp, vgpr in _kernel_notes(o, str(tmp_path)):
            total += 1
            assert scratch == 0 and vsp == 0, "%s: %d bytes of scratch, %d VGPR spills (vgpr_count %d)" % (name, scratch, vsp, vgpr)
    assert total >= 40   # pre/post kernels + every conv instantiation


# ---- hazard rules of DESIGN.md section 5 as build-time checks (tools/isa_check.py: CFG dataflow over the disassembly) ----
import sys  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_check  # noqa: E402

_needs_llvm = pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="LLVM tools not installed")


@pytest.fixture(scope="module")
def disasm(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("isa"))
    return {f: isa_check.disassemble(os.path.join(BUILD, f + ".o"), tmp) for f in ("kernels_pre", "kernels_post", "kernels_conv", "kernels_ring", "kernels_f32")}


@_needs_llvm
def test_isa_checker_detects_the_patterns_it_guards(tmp_path):
    """The analysis itself, on hand-written instruction lists: the round-2 pw_stream shape (a ds_read sunk below its
    MFMAs, bare barrier, DMA into the slot) is reported, the repaired shape is not; a 128-bit LDS read between
    s_and_saveexec and the restoring s_or is reported, after the restore it is not."""
    I = lambda a, m, o="", t=None: (a, m, o, t)
    racy = [I(0, "ds_read_b128", "v[0:3], v9"), I(4, "v_mfma_f32_16x16x32_bf16", "a[0:3], v[4:7], v[8:11], a[0:3]"),
            I(8, "s_barrier"), I(12, "s_waitcnt", "lgkmcnt(0)"), I(16, "buffer_load_dwordx4", "v4, s[4:7], 0 offen lds"), I(20, "s_endpgm")]
    assert isa_check.uses_lds_dma(racy) and isa_check.pending_lds_reads_at_barriers(racy) == ["0x8"]
    fixed = racy[:2] + [I(6, "s_waitcnt", "lgkmcnt(0)")] + racy[2:]
    assert isa_check.pending_lds_reads_at_barriers(fixed) == []
    loop = [I(0, "s_waitcnt", "lgkmcnt(0)"), I(4, "s_barrier"), I(8, "ds_read_b64", "v[0:1], v9"), I(12, "s_cbranch_scc1", "65533", 0), I(16, "s_endpgm")]
    assert isa_check.pending_lds_reads_at_barriers(loop) == []          # the wait at the loop head covers the back edge
    bare = [I(4, "s_barrier"), I(8, "ds_read_b64", "v[0:1], v9"), I(12, "s_cbranch_scc1", "65533", 4), I(16, "s_endpgm")]
    assert isa_check.pending_lds_reads_at_barriers(bare) == ["0x4"]      # without it the read of iteration i meets the barrier of i + 1
    masked = [I(0, "s_and_saveexec_b64", "s[0:1], vcc"), I(4, "ds_read_b128", "v[0:3], v9"), I(8, "s_or_b64", "exec, exec, s[0:1]"),
              I(12, "ds_read_b128", "v[4:7], v9"), I(16, "s_endpgm")]
    assert isa_check.ds_read_b128_under_partial_exec(masked) == ["0x4"]


@_needs_llvm
def test_no_lds_read_in_flight_at_a_barrier_of_an_lds_dma_kernel(disasm):
    """Every kernel that stages operands by LDS-DMA re-fills ring slots right behind its barriers; an LDS read still in
    flight at such a barrier races with the DMA into its slot.  Found twice in shipped ISA: pw_stream (round 2, hipcc sank
    the last ds_read pair of a K step below the bare s_barrier) and conv3x3_kx (round 3, same shape)."""
    n = 0
    for f, kernels in disasm.items():
        for name, ins in kernels.items():
            if not isa_check.uses_lds_dma(ins):
                continue
            n += 1
            bad = isa_check.pending_lds_reads_at_barriers(ins)
            assert not bad, "%s: LDS reads may be in flight at s_barrier %s (write `s_waitcnt lgkmcnt(0)` in front of it)" % (name, bad)
    assert n >= 33


@_needs_llvm
def test_counted_vmcnt_publishes_only_from_waves_without_stores(disasm):
    """Round 4: the ring convolutions (kernels_ring.hip) publish LDS-DMA tiles behind COUNTED `s_waitcnt vmcnt(N)` -- sound only
    if every operation the count leaves outstanding is a later LDS-DMA, i.e. the publishing wave never has a vector store or
    atomic in flight.  The checker itself is exercised on hand-written lists; then every kernel of the library is held to the
    rule, and the ring kernels must actually contain publishing counted waits (so the rule is not vacuous) while every other
    LDS-DMA kernel still drains."""
    I = lambda a, m, o="", t=None: (a, m, o, t)
    dma = I(0, "buffer_load_dwordx4", "v4, s[4:7], 0 offen lds")
    ok = [dma, I(8, "buffer_load_dwordx4", "v5, s[4:7], 0 offen lds"), I(16, "s_waitcnt", "vmcnt(1)"), I(20, "ds_add_u32", "v1, v0"), I(24, "s_endpgm")]
    assert isa_check.counted_vmcnt_with_store_in_flight(ok) == [] and isa_check.counted_vmcnt_waits(ok) == ["0x10"]
    bad = [dma, I(8, "global_store_dwordx4", "v[2:3], v[6:9], off"), I(16, "s_waitcnt", "vmcnt(1)"), I(20, "ds_add_u32", "v1, v0"), I(24, "s_endpgm")]
    assert isa_check.counted_vmcnt_with_store_in_flight(bad) == ["0x10"]
    drained = [dma, I(8, "global_store_dwordx4", "v[2:3], v[6:9], off"), I(12, "s_waitcnt", "vmcnt(0)"), I(14, "buffer_load_dwordx4", "v4, s[4:7], 0 offen lds"),
               I(16, "s_waitcnt", "vmcnt(1)"), I(20, "s_barrier"), I(24, "s_endpgm")]
    assert isa_check.counted_vmcnt_with_store_in_flight(drained) == []       # the store retired before the counted wait
    epilogue = [dma, I(8, "global_store_dwordx4", "v[2:3], v[6:9], off"), I(16, "s_waitcnt", "vmcnt(1)"), I(20, "v_add_f32", "v0, v1, v2"), I(24, "s_endpgm")]
    assert isa_check.counted_vmcnt_with_store_in_flight(epilogue) == []      # a register-use wait publishes nothing
    ring = 0
    for f, kernels in disasm.items():
        for name, ins in kernels.items():
            bad = isa_check.counted_vmcnt_with_store_in_flight(ins)
            assert not bad, "%s: a counted vmcnt publishes LDS-DMA data at %s while a store of the same wave may be in flight" % (name, bad)
            n = len(isa_check.counted_vmcnt_waits(ins))
            if "conv_ring_kernel" in name:
                assert n >= 1, "%s: no publishing counted wait found (the rule would be vacuous)" % name
                ring += 1
            else:
                assert n == 0, "%s: a counted vmcnt publishes LDS-DMA data outside the ring kernels (every other kernel drains)" % name
    assert ring == 3


@_needs_llvm
def test_no_128_bit_lds_read_in_pre_and_post_kernels(disasm):
    """Rules (ii)/(iii): the short pre/post kernels (preprocess, align, decode, sort, NMS, selection) co-reside with the other
    chain's MFMA waves and keep their LDS operands to 64-bit reads / v_readlane."""
    for f in ("kernels_pre", "kernels_post"):
        assert len(disasm[f]) >= 4
        for name, ins in disasm[f].items():
            assert not isa_check.has_instr(ins, "ds_read_b128"), "%s uses ds_read_b128" % name


@_needs_llvm
def test_no_128_bit_lds_read_under_partial_exec(disasm):
    """Rule (i): a ds_read_b128 never executes at a point where EXEC may be narrowed (divergent branch / loop).  Round 3 found
    the compiler sinking the stem's pooling reads under the store's predicate although the source predicated only the store."""
    for f, kernels in disasm.items():
        for name, ins in kernels.items():
            bad = isa_check.ds_read_b128_under_partial_exec(ins)
            assert not bad, "%s: ds_read_b128 under a possibly partial EXEC at %s" % (name, bad[:8])


@_needs_llvm
def test_every_persistent_kernel_requests_the_whole_cu(disasm, tmp_path):
    """Rule 2 / (iv): a persistent workgroup owns its CU's LDS.  The library lists its persistent kernels with the dynamic LDS
    their (single, shared) launch path requests; this walks the list against the code object: every entry asks for 160 KiB,
    every entry names kernels that exist, and every 8-wave kernel with dynamic LDS that stages by LDS-DMA is either in the
    list or one of the one-tile-per-workgroup kernels named here -- so a new persistent kernel cannot skip the rule."""
    import ctypes as C
    import rfd_hip
    L = rfd_hip.load_library()
    n = L.rfd_debug_persistent_kernel(-1, None, None)
    assert n >= 7
    table = {}
    for i in range(n):
        name, lds = C.c_char_p(), C.c_size_t()
        assert L.rfd_debug_persistent_kernel(i, C.byref(name), C.byref(lds)) == n
        table[name.value.decode()] = lds.value
    assert all(v == 160 * 1024 for v in table.values()), table
    notes = {k[0]: k for k in _kernel_notes(os.path.join(BUILD, "kernels_conv.o"), str(tmp_path))}
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(str(tmp_path), "dev.co")], check=True, capture_output=True, text=True).stdout
    wg = {}
    for block in txt.split("- .agpr_count:")[1:]:
        nm = re.search(r"\.name:\s+(\S+)", block).group(1)
        wg[nm] = (int(re.search(r"\.max_flat_workgroup_size:\s+(\d+)", block).group(1)), int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", block).group(1)))
    one_tile_per_workgroup = ("conv_b2b_s1_kernel", "conv_igemm_kernel", "conv3x3_kx_kernel")   # 8-wave, LDS-DMA, NOT persistent: sized to co-reside
    seen = set()
    for name, ins in disasm["kernels_conv"].items():
        if wg[name][0] != 512 or not isa_check.uses_lds_dma(ins):
            continue
        hit = [t for t in table if ("3rfd%d%sE" % (len(t), t)) in name or ("3rfd%d%sI" % (len(t), t)) in name]
        if hit:
            seen.add(hit[0])
            assert wg[name][1] == 0, "%s: static LDS next to the 160 KiB dynamic request" % name
        else:
            assert any(("%d%s" % (len(t), t)) in name for t in one_tile_per_workgroup), "%s: 8-wave LDS-DMA kernel outside the persistent launch path" % name
    assert seen == set(table), "listed but not in the code object: %s" % (set(table) - seen)

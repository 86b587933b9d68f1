"""CPU tests of the drop-in boundary: librfd_hip.so loads without a GPU, exports every symbol that
include/rfd.h declares, describes the build-defined graph, and refuses to run without a device."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "rfd.h")).read()
    return sorted(set(re.findall(r"RFD_API\s+[\w \*]+?\b(rfd_\w+|_nms)\s*\(", txt)))


def test_header_symbols_exported(rfd):
    declared = _declared()
    assert len(declared) >= 30
    L = rfd.load_library()
    for name in declared:
        assert hasattr(L, name), "librfd_hip.so does not export %s" % name
    assert sorted(rfd.API_SYMBOLS) == declared


def test_exports_only_the_c_abi(rfd):
    out = subprocess.check_output(["nm", "-D", "--defined-only", rfd.LIB_PATH]).decode()
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert sorted(syms) == _declared()  # no C++ symbols leak; plain C names only


def test_graph_matches_survey_appendix_b(rfd):
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    assert abs(g.macs - 44.2646528e9) < 1e3          # 44.265 GMAC / image
    convs = [o for o in g.ops if o.kind in (0, 2, 3, 6)]
    heads = [o for o in g.ops if o.head_softmax]
    fused_sc = [o for o in g.ops if o.layer2 >= 0]     # 4 shortcut convs ride in their conv3's GEMM
    fused_n = [o for o in g.ops if o.layer_n2 >= 0]    # 6 SSH sibling pairs run as one GEMM along N
    b2b = [o for o in g.ops if o.kind == 6]             # conv3 + next conv1 back to back (stage 1; since round 3 stage 2's dim-match units)
    assert len(convs) + 2 * len(heads) + len(fused_sc) + len(fused_n) + len(b2b) == 82  # 9 head convs = 3 fused N=32 GEMMs
    assert len(b2b) == 11
    assert sorted(g.layers[o.layer].name.decode() for o in b2b) == ["stage1_unit1_conv3", "stage1_unit2_conv3", "stage1_unit3_conv3",
                                                                   "stage2_unit1_conv3", "stage2_unit2_conv3", "stage2_unit3_conv3", "stage2_unit4_conv3", "stage3_unit2_conv3",
                                                                   "stage3_unit3_conv3", "stage3_unit4_conv3", "stage3_unit5_conv3"]
    assert len(fused_n) == 6
    assert len(fused_sc) == 4 and g.num_layers == 76
    assert sorted(t.head_level for t in g.tensors if t.head_level) == [1, 2, 3]
    hl = {t.head_level: (t.height, t.width, t.channels) for t in g.tensors if t.head_level}
    assert hl == {1: (20, 20, 32), 2: (40, 40, 32), 3: (80, 80, 32)}
    params = sum(l.cout * l.kh * l.kw * l.cin for l in g.layers)
    assert abs(params - 27.24e6) < 0.05e6              # 27.24 M conv parameters
    # 1920x1088 input: 225.75 GMAC (Appendix B)
    g2 = rfd.Graph(rfd.BACKBONE_R50, 1920, 1088)
    assert abs(g2.macs / 1e9 - 225.75) < 0.01


def test_graph_plan_has_no_aliasing(rfd):
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    # tensors that share a buffer must have disjoint live ranges: an op never reads and writes one buffer
    for o in g.ops:
        outs = [t for t in (o.out, o.out2, o.outf, o.out_b) if t >= 0]
        ins = [t for t in (o.in_, o.in2, o.res) if t >= 0]
        ob = [g.tensors[t].buffer for t in outs]
        assert len(set(ob)) == len(ob)
        if o.out == o.in_:  # SSH: reads one channel slice of the concat buffer, writes a disjoint one
            L = g.layers[o.layer]
            nout = L.cout + (g.layers[o.layer_n2].cout if o.layer_n2 >= 0 else 0)
            rd = set(range(o.x_coff, o.x_coff + L.cin))
            wr = {o.y_coff + n + (o.y_split_add if n >= o.y_split else 0) for n in range(nout)}
            assert not rd & wr
            continue
        assert not set(ob) & {g.tensors[t].buffer for t in ins}


def test_mobilenet_graph(rfd):
    g = rfd.Graph(rfd.BACKBONE_MNET025, 640, 640)           # BASELINE.json configs[1]
    assert sum(1 for l in g.layers if l.kind == 1) == 13    # depthwise 3x3 layers
    assert abs(g.macs - 0.9811456e9) < 1e3
    hl = {t.head_level: (t.height, t.width, t.channels) for t in g.tensors if t.head_level}
    assert hl == {1: (20, 20, 32), 2: (40, 40, 32), 3: (80, 80, 32)}   # same head / anchor contract as R50
    assert all(t.channels % 64 == 0 or t.is_f32 or t.is_input for t in g.tensors)


def test_bad_graph_arguments(rfd):
    with pytest.raises(rfd.RfdError) as e:
        rfd.Graph(rfd.BACKBONE_R50, 641, 640)
    assert e.value.status == rfd.RFD_ERR_INVALID_ARG
    with pytest.raises(rfd.RfdError):
        rfd.Graph(7, 640, 640)


def test_missing_library_fails_loudly(rfd):
    with pytest.raises(ImportError):
        rfd.load_library("/nonexistent/librfd_hip.so")


def test_no_device_no_fallback(rfd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rfd.RfdError) as e:
        rfd.RetinaFaceDetection()
    assert e.value.status == rfd.RFD_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)
    # the reference-compatible `_nms` reports failure through num_out = -1 instead of printing
    L = rfd.load_library()
    keep = (ctypes.c_int32 * 4)()
    num = ctypes.c_int(0)
    boxes = (ctypes.c_float * 20)()
    L._nms(keep, ctypes.byref(num), boxes, 4, 5, 0.4, 0)
    assert num.value == -1


def test_config_layout_and_precision_validation(rfd):
    """rfd_config keeps its round-1 size (the `precision` field took the first reserved word) and rfd_create validates it before
    it looks for a device, so the check runs here without a GPU."""
    cfg = rfd.rfd_config()
    assert ctypes.sizeof(cfg) == 16 * 4
    assert rfd.rfd_config.precision.offset == 10 * 4 and rfd.rfd_config.backbone.offset == 9 * 4
    L = rfd.load_library()
    L.rfd_config_default(ctypes.byref(cfg))
    assert cfg.precision == rfd.PRECISION_BF16 and cfg.image_w == 640 and cfg.max_det == 1024
    ctx = ctypes.c_void_p()
    cfg.precision = 7
    assert L.rfd_create(ctypes.byref(cfg), ctypes.byref(ctx)) == rfd.RFD_ERR_INVALID_ARG
    cfg.precision = rfd.PRECISION_F32
    cfg.backbone = rfd.BACKBONE_MNET025
    assert L.rfd_create(ctypes.byref(cfg), ctypes.byref(ctx)) == rfd.RFD_ERR_INVALID_ARG
    assert b"R50" in L.rfd_last_error()

"""The persistent kernels (pw_stream_kernel: short-K wide-N pointwise layers; pw_gemm_kernel: long-K pointwise layers; pw_wide_kernel: wide pointwise GEMMs;
conv3x3_c64_kernel: 64 -> 64 3x3 with the filter bank resident in LDS) against the generic implicit-GEMM kernel on the SAME inputs at batch sizes where every
workgroup walks SEVERAL tiles (the per-op tests of test_network_gpu.py run at n = 2: one tile per workgroup) and where the
last tile is partial.  Both paths accumulate in the same K order, so the outputs must be bit-identical; the generic
kernel itself is checked against torch-CPU f32 in test_network_gpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [32, 17])
def test_persistent_kernels_equal_generic_kernels_bitwise(rfd, n):
    det = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=32, max_det=16)
    det.init_synthetic_weights(1234)
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    rng = np.random.default_rng(n)
    ops = []
    wide = set()
    s1pair = set()
    for i, o in enumerate(g.ops):
        L = g.layers[o.layer]
        if o.kind == 6:   # stage-1 back-to-back kernel, with and without the fused shortcut: persistent forms (tile 6)
            ops.append(i)
            if L.cin == 64 and g.layers[o.layer_b].cout == 64:   # ... and the weight-resident pair kernel (tile 16; round 4)
                s1pair.add(i)
            continue
        if o.kind != 2:
            continue
        pw = L.kh == 1 and o.res >= 0 and o.layer2 < 0 and L.cin in (64, 128, 256) and L.cout >= 4 * L.cin
        c64 = L.kh == 3 and L.stride == 1 and L.cin == 64 and L.cout == 64 and o.layer_n2 < 0
        # pw_gemm_kernel: long-K pointwise layers without a residual (conv1 of the units)
        # (with or without a residual: the FPN laterals add the coarser level, nearest-2x upsampled, after the ReLU)
        pwg = L.kh == 1 and L.stride == 1 and o.layer2 < 0 and o.out2 < 0 and o.outf < 0 \
            and o.layer_b < 0 and L.cin % 128 == 0 and L.cin >= 256 and L.cout % 128 == 0 and L.cout <= 1024
        # pw_wide_kernel: wide pointwise GEMMs (conv3 + fused stride-2 shortcut of the down-sampling units, stage-4 conv3)
        kk = L.cin + (g.layers[o.layer2].cin if o.layer2 >= 0 else 0)
        pww = L.kh == 1 and L.stride == 1 and o.in_affine < 0 and o.outf < 0 and o.layer_b < 0 and kk >= 384 \
            and L.cout % 256 == 0 and 512 <= L.cout <= 2048 and not o.res_up2 and not o.res_post
        if pw or c64 or pwg or pww:
            ops.append(i)
            wide.add(i) if pww else None
    assert len(ops) >= 22 and len(wide) >= 5 and len(s1pair) == 2
    checked = 0
    for i in ops:
        o = g.ops[i]
        for t in (o.in_, o.res, o.in2):
            if t < 0:
                continue
            td = g.tensors[t]
            x = rng.normal(0, 1, size=(n, td.height, td.width, td.channels)).astype(np.float32)
            if t == o.in_:
                x = np.maximum(x, 0)
            det.debug_write(t, (x.view(np.uint32) >> 16).astype(np.uint16))
        outs = [t for t in (o.out, o.out2, o.out_b) if t >= 0]
        res = {}
        # 7: generic kernels only; 6: persistent kernels forced whatever the problem size (the first that accepts the layer);
        # 12: pw_wide_kernel forced (layers that pw_stream / pw_gemm would take first)
        for tile in ((7, 6, 12) if i in wide else (7, 6, 16) if i in s1pair else (7, 6)):
            det.debug_set_conv_tile(tile)
            if o.out >= 0 and o.out == o.in_:   # SSH: the op writes a channel slice of its own input tensor
                pass
            for rep in range(2 if tile != 7 else 1):   # twice: the persistent path must also be repeatable
                for t in outs:
                    if t != o.in_:
                        td = g.tensors[t]
                        det.debug_write(t, np.full((n, td.height, td.width, td.channels), 0x7fc0, np.uint16))  # NaN poison
                det.debug_run(n, i, i)
                got = [det.debug_read(t, n, g.tensors[t]) for t in outs]
                if tile != 7 and rep == 1:
                    for a, b in zip(got, res[tile]):
                        assert np.array_equal(a, b), "op %d tile %d: persistent kernel not repeatable" % (i, tile)
                res[tile] = got
        for tile in [t for t in res if t != 7]:
            for t, a, b in zip(outs, res[tile], res[7]):
                if t == o.in_:   # in-place slice writers: compare only the written channels
                    L = g.layers[o.layer]
                    a, b = a[..., o.y_coff:o.y_coff + L.cout], b[..., o.y_coff:o.y_coff + L.cout]
                bad = int((a != b).sum())
                assert bad == 0, "op %d (%s) tile %d tensor %d: %d / %d elements differ from the generic kernel at n = %d" % (
                    i, g.layers[o.layer].name.decode(), tile, t, bad, a.size, n)
                checked += 1
    det.debug_set_conv_tile(0)
    det.close()
    assert checked >= 10


def _bf16_to_f32(a):
    return (a.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("n", [16, 5])
def test_halo_kernel_equals_merged_kx_kernel_bitwise(rfd, n):
    """conv3x3_halo_kernel (Cin >= 128 3x3 layers; force_tile 13: 128-channel items, 14: 256-channel items) against the
    merged-kx kernel (7).  Both accumulate in the order (chunk, ky, kx, 32-wide MFMA step), so whichever of them the
    size heuristic of launch_conv picks -- it depends on the batch size -- the layer's output is bit-identical: a frame's
    detections do not depend on the batch it arrives in.  Several items per workgroup (n = 16), partial last tiles (40 x 40
    maps), item counts below the CU count (n = 5); every variant must also be bit-repeatable."""
    det = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=16, max_det=16)
    det.init_synthetic_weights(4321)
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    rng = np.random.default_rng(100 + n)
    ops = []
    for i, o in enumerate(g.ops):
        L = g.layers[o.layer]
        # (layer_n2 >= 0: the SSH conv1 + ctx1 pair, one 192-channel GEMM with two destinations -> 192-channel items)
        if o.kind == 2 and L.kh == 3 and L.stride == 1 and L.cin % 128 == 0 and L.cout % 128 == 0 and L.cout <= 512 \
                and o.res < 0 and g.tensors[o.in_].width in (80, 40):
            ops.append(i)
    assert len(ops) >= 10 and any(g.ops[i].layer_n2 >= 0 for i in ops)
    checked = 0
    for i in ops:
        o = g.ops[i]
        L = g.layers[o.layer]
        td = g.tensors[o.in_]
        x = np.maximum(rng.normal(0, 1, size=(n, td.height, td.width, td.channels)), 0).astype(np.float32)
        det.debug_write(o.in_, (x.view(np.uint32) >> 16).astype(np.uint16))
        res = {}
        tiles = (7, 13) + ((14,) if L.cout % 256 == 0 and td.width != 40 and o.layer_n2 < 0 else ())
        for tile in tiles:
            det.debug_set_conv_tile(tile)
            for rep in range(1 if tile == 7 else 2):
                to = g.tensors[o.out]
                det.debug_write(o.out, np.full((n, to.height, to.width, to.channels), 0x7fc0, np.uint16))
                det.debug_run(n, i, i)
                got = det.debug_read(o.out, n, to)
                if rep == 1:
                    assert np.array_equal(got, res[tile]), "op %d tile %d: not repeatable" % (i, tile)
                res[tile] = got
        want = res[7]                                   # whole tensor: channels the op does not write keep the poison in both
        assert not (want[..., o.y_coff:o.y_coff + L.cout] == 0x7fc0).any()
        for tile in tiles[1:]:
            got = res[tile]
            bad = int((got != want).sum())
            assert bad == 0, "op %d (%s) tile %d: %d / %d elements differ from the merged-kx kernel at n = %d" % (
                i, L.name.decode(), tile, bad, got.size, n)
            checked += 1
    det.debug_set_conv_tile(0)
    det.close()
    assert checked >= 12


@pytest.mark.parametrize("size", [(768, 480), (640, 448)])
def test_forced_persistent_kernels_on_other_input_sizes(rfd, size):
    """Feature maps that do not divide into the persistent kernels' items: 96 x 60 / 48 x 30 (16 x 16 halo tiles with a partial
    last tile row, at 768 x 480) and 80 x 56 / 40 x 28 (the 40-wide row tile with 4 of 6 rows in its last item, at 640 x 448),
    pixel counts that are not multiples of 256, n = 3.  Every convolution op, persistent kernels forced (6) against the
    generic / merged-kx kernels (7): bit-identical, NaN-poisoned outputs fully overwritten where the op writes."""
    n = 3
    det = rfd.RetinaFaceDetection(image_size=size, max_batch_size=n, max_det=16)
    det.init_synthetic_weights(77)
    g = rfd.Graph(rfd.BACKBONE_R50, size[0], size[1])
    rng = np.random.default_rng(size[0])
    checked = 0
    for i, o in enumerate(g.ops):
        if o.kind not in (2, 6):
            continue
        L = g.layers[o.layer]
        for t in (o.in_, o.res, o.in2):
            if t < 0:
                continue
            td = g.tensors[t]
            x = rng.normal(0, 1, size=(n, td.height, td.width, td.channels)).astype(np.float32)
            if t == o.in_:
                x = np.maximum(x, 0)
            det.debug_write(t, (x.view(np.uint32) >> 16).astype(np.uint16))
        outs = [t for t in (o.out, o.out2, o.out_b) if t >= 0 and not g.tensors[t].is_f32]
        res = {}
        for tile in (7, 6):
            det.debug_set_conv_tile(tile)
            for t in outs:
                if t != o.in_:
                    td = g.tensors[t]
                    det.debug_write(t, np.full((n, td.height, td.width, td.channels), 0x7fc0, np.uint16))
            det.debug_run(n, i, i)
            res[tile] = [det.debug_read(t, n, g.tensors[t]) for t in outs]
        for t, a, b in zip(outs, res[6], res[7]):
            if t == o.in_:
                a, b = a[..., o.y_coff:o.y_coff + L.cout], b[..., o.y_coff:o.y_coff + L.cout]
            assert np.array_equal(a, b), "op %d (%s) tensor %d at %s" % (i, L.name.decode(), t, size)
            checked += 1
    det.debug_set_conv_tile(0)
    det.close()
    assert checked >= 60


@pytest.mark.parametrize("size", [(640, 640), (480, 352)])
def test_persistent_stem_equals_one_tile_stem_bitwise(rfd, size):
    """The fused stem (conv0 + max pool + BN1) runs as a persistent kernel from 4 tiles per workgroup slot (round 4: weights stay in
    registers, the next tile's input patch is prefetched under the conv phase) and as one workgroup per tile below that.  Same
    arithmetic per tile: 16 frames in one launch (persistent: 6 400 tiles) against the same frames two at a time (800 tiles: the
    one-tile kernel), and a batch whose last workgroup gets a short run of tiles."""
    nb = 16 if size == (640, 640) else 40      # (480 x 352: 30 x 22 pooled pixels per image -> partial tiles on both edges)
    det = rfd.RetinaFaceDetection(image_size=size, max_batch_size=nb, max_det=16)
    det.init_synthetic_weights(1234)
    g = rfd.Graph(rfd.BACKBONE_R50, size[0], size[1])
    op = next(i for i, o in enumerate(g.ops) if o.kind == 3)
    o = g.ops[op]
    tin, tout = g.tensors[o.in_], g.tensors[o.out]
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, size=(nb, tin.height, tin.width, tin.channels)).astype(np.float32)
    x[..., 3] = 0
    bits = (x.view(np.uint32) >> 16).astype(np.uint16)        # small integers are exact in bf16
    # (introspection answers for a real pass, where the conv behind the stem rides in the persistent kernel: <true>)
    assert det.debug_op_kernels(nb, op) == ["stem_persistent_kernel<true>"] and det.debug_op_kernels(2, op) == ["stem_kernel"]
    want = []
    for i in range(0, nb, 2):
        det.debug_write(o.in_, bits[i:i + 2])
        det.debug_run(2, op, op)
        want.append(det.debug_read(o.out, 2, tout))
    want = np.concatenate(want)
    for n in ((16, 7) if nb == 16 else (40, 37)):
        det.debug_write(o.in_, bits[:n])
        det.debug_write(o.out, np.full((n, tout.height, tout.width, tout.channels), 0x7fc0, np.uint16))
        det.debug_run(n, op, op)
        got = det.debug_read(o.out, n, tout)
        assert np.array_equal(got, want[:n]), "n = %d: %d elements differ" % (n, int((got != want[:n]).sum()))
    det.close()


@pytest.mark.parametrize("size", [(640, 640), (480, 352)])
def test_stem_with_the_first_conv1_fused_equals_the_two_ops_bitwise(rfd, size):
    """Round 4: when a pass runs the stem and the first unit's conv1 (1x1, 64 -> 64, bias + ReLU) back to back, the persistent stem
    kernel computes the conv on its pooled tile (the pooling pass's lane layout IS the MFMA's B-fragment layout) and the conv's own
    launch is skipped.  Against the two ops run one at a time (the stem alone, then the generic conv kernel on the stored output):
    both tensors bit-identical, including partial tiles on both edges (480 x 352) and a short last run of tiles."""
    nb = 16 if size == (640, 640) else 40
    det = rfd.RetinaFaceDetection(image_size=size, max_batch_size=nb, max_det=16)
    det.init_synthetic_weights(1234)
    g = rfd.Graph(rfd.BACKBONE_R50, size[0], size[1])
    op = next(i for i, o in enumerate(g.ops) if o.kind == 3)
    o, o1 = g.ops[op], g.ops[op + 1]
    L1 = g.layers[o1.layer]
    assert o1.kind == 2 and o1.in_ == o.out and (L1.kh, L1.cin, L1.cout) == (1, 64, 64)
    tin, ty, tt = g.tensors[o.in_], g.tensors[o.out], g.tensors[o1.out]
    rng = np.random.default_rng(4)
    x = rng.integers(0, 256, size=(nb, tin.height, tin.width, tin.channels)).astype(np.float32)
    x[..., 3] = 0
    bits = (x.view(np.uint32) >> 16).astype(np.uint16)
    poison = lambda n, t: np.full((n, t.height, t.width, t.channels), 0x7fc0, np.uint16)
    assert det.debug_op_kernels(nb, op + 1) == ["(fused into stem_persistent_kernel<true>)"]
    assert det.debug_op_kernels(2, op + 1)[0].startswith("conv_igemm_kernel")      # below the persistent stem's threshold: its own launch
    for n in ((16, 7) if nb == 16 else (40, 37)):
        det.debug_write(o.in_, bits[:n])
        det.debug_write(o.out, poison(n, ty)); det.debug_write(o1.out, poison(n, tt))
        det.debug_run(n, op, op)            # one op at a time: nothing to fuse
        det.debug_run(n, op + 1, op + 1)
        want_y, want_t = det.debug_read(o.out, n, ty), det.debug_read(o1.out, n, tt)
        det.debug_write(o.out, poison(n, ty)); det.debug_write(o1.out, poison(n, tt))
        det.debug_run(n, op, op + 1)        # the range: the fused kernel
        got_y, got_t = det.debug_read(o.out, n, ty), det.debug_read(o1.out, n, tt)
        assert np.array_equal(got_y, want_y), "n = %d: stem output differs in %d elements" % (n, int((got_y != want_y).sum()))
        assert np.array_equal(got_t, want_t), "n = %d: conv1 output differs in %d elements" % (n, int((got_t != want_t).sum()))
        assert not (got_t == 0x7fc0).any()
    det.close()

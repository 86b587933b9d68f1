"""The wave-specialised ring convolution (kernels_ring.hip: 4 loader waves + 4 MFMA consumer waves over an LDS ring, counted
vmcnt waits in the loader waves only) against the barrier-per-step kernels it replaces, on the SAME inputs: both accumulate
in the same K order with the same MFMA sequence, so every output must be bit-identical.  Batch sizes: 16 (the production
chain), 5 (M = 2000 / 8000: partial last tiles, fewer workgroups than CUs) and 1.  The barrier-per-step kernels themselves
are checked against torch-CPU f32 in test_network_gpu.py (which also runs every op in ring mode, tile 17)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RING = 17      # force_tile: the ring wherever the layer shape allows
GENERIC = 7    # force_tile: one-tile-per-workgroup kernels only (conv_igemm / conv3x3_kx)


def _ring_ops(g):
    ops = []
    for i, o in enumerate(g.ops):
        if o.kind != 2 or o.layer_b >= 0:
            continue
        L = g.layers[o.layer]
        cout = L.cout + (g.layers[o.layer_n2].cout if o.layer_n2 >= 0 else 0)
        if cout % 128 == 0 and L.cin % 64 == 0:
            ops.append(i)
    return ops


def _tile_ops(g):
    """... and the fused SSH pairs (Cout = 192: the 128 x 192 tile, which the ring does not take)"""
    extra = [i for i, o in enumerate(g.ops) if o.kind == 2 and o.layer_b < 0 and o.layer_n2 >= 0
             and g.layers[o.layer].cout + g.layers[o.layer_n2].cout == 192]
    return sorted(set(_ring_ops(g)) | set(extra))


@pytest.mark.parametrize("n", [16, 5, 1])
def test_ring_kernels_equal_barrier_kernels_bitwise(rfd, n):
    det = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=16, max_det=16)
    det.init_synthetic_weights(1234)
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    rng = np.random.default_rng(100 + n)
    ops = _ring_ops(g)
    assert len(ops) >= 30
    kinds = set()
    for i in ops:
        o = g.ops[i]
        L = g.layers[o.layer]
        for t in (o.in_, o.res, o.in2):
            if t < 0:
                continue
            td = g.tensors[t]
            x = rng.normal(0, 1, size=(n, td.height, td.width, td.channels)).astype(np.float32)
            if t == o.in_ and o.in_affine < 0:
                x = np.maximum(x, 0)
            det.debug_write(t, (x.view(np.uint32) >> 16).astype(np.uint16))
        outs = [t for t in (o.out, o.out2) if t >= 0]
        res = {}
        for tile in (GENERIC, RING):
            det.debug_set_conv_tile(tile)
            for rep in range(2 if tile == RING else 1):   # twice: the ring's flag protocol must be repeatable
                for t in outs:
                    if t != o.in_:
                        td = g.tensors[t]
                        det.debug_write(t, np.full((n, td.height, td.width, td.channels), 0x7fc0, np.uint16))  # NaN poison
                det.debug_run(n, i, i)
                got = [det.debug_read(t, n, g.tensors[t]) for t in outs]
                if rep == 1:
                    for a, b in zip(got, res[tile]):
                        assert np.array_equal(a, b), "op %d: ring kernel not repeatable" % i
                res[tile] = got
        for t, a, b in zip(outs, res[RING], res[GENERIC]):
            if t == o.in_:   # in-place slice writers (SSH concat): compare only the written channels
                a, b = a[..., o.y_coff:o.y_coff + L.cout], b[..., o.y_coff:o.y_coff + L.cout]
            bad = int((a != b).sum())
            assert bad == 0, "op %d (%s k%d %d->%d) tensor %d: %d / %d elements differ from the barrier-per-step kernel at n = %d" % (
                i, L.name.decode(), L.kh, L.cin, L.cout, t, bad, a.size, n)
        kinds.add((L.kh, L.stride, o.in_affine >= 0, o.layer2 >= 0, o.res >= 0))
    det.debug_set_conv_tile(0)
    # the sweep covers 1x1 and 3x3, stride 1 and 2, operand affine, fused shortcut segment, residual
    assert {k[0] for k in kinds} == {1, 3} and {k[1] for k in kinds} == {1, 2}
    assert any(k[2] for k in kinds) and any(k[3] for k in kinds) and any(k[4] for k in kinds)
    det.close()


@pytest.mark.parametrize("n", [16, 3])
def test_eight_wave_tiles_equal_four_wave_tiles_bitwise(rfd, n):
    """Round 4: the one-tile-per-workgroup kernels run the 128 x 128 tile with EIGHT waves (32 x 64 wave tiles: two waves per SIMD
    from one workgroup) where the grid gives a CU a single workgroup -- conv3x3_kx_kernel<128, 4, 2> and
    conv_igemm_kernel<128, 128, 4, 2, 3>, what force_tile 7 (and 0 at small sizes) now selects.  Same K order, same MFMA sequence per
    output: bit-identical to the four-wave forms (force_tile 19: the four-wave merged-kx kernel; force_tile 1: the four-wave
    generic tile for every conv)."""
    det = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=16, max_det=16)
    det.init_synthetic_weights(1234)
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    rng = np.random.default_rng(200 + n)
    ops = _tile_ops(g)
    names = set()
    for i in ops:
        o = g.ops[i]
        L = g.layers[o.layer]
        det.debug_set_conv_tile(GENERIC)
        names.update(det.debug_op_kernels(n, i))
        for t in (o.in_, o.res, o.in2):
            if t < 0:
                continue
            td = g.tensors[t]
            x = rng.normal(0, 1, size=(n, td.height, td.width, td.channels)).astype(np.float32)
            if t == o.in_ and o.in_affine < 0:
                x = np.maximum(x, 0)
            det.debug_write(t, (x.view(np.uint32) >> 16).astype(np.uint16))
        outs = [t for t in (o.out, o.out2) if t >= 0]
        res = {}
        # (a 3x3 stride-1 layer that the halo kernel does not take -- W = 20 -- accumulates chunk-major in the merged-kx kernel and
        #  tap-major in the generic one: tile 1 is not its four-wave form, tile 19 is)
        kx_layer = L.kh == 3 and L.stride == 1
        tiles = (GENERIC, 19) if kx_layer else (GENERIC, 1)
        for tile in tiles:
            det.debug_set_conv_tile(tile)
            for t in outs:
                if t != o.in_:
                    td = g.tensors[t]
                    det.debug_write(t, np.full((n, td.height, td.width, td.channels), 0x7fc0, np.uint16))  # NaN poison
            det.debug_run(n, i, i)
            res[tile] = [det.debug_read(t, n, g.tensors[t]) for t in outs]
        for tile in tiles[1:]:
            for t, a, b in zip(outs, res[tile], res[GENERIC]):
                if t == o.in_:
                    a, b = a[..., o.y_coff:o.y_coff + L.cout], b[..., o.y_coff:o.y_coff + L.cout]
                bad = int((a != b).sum())
                assert bad == 0, "op %d (%s k%d %d->%d) tensor %d: tile %d differs from the eight-wave form in %d / %d elements at n = %d" % (
                    i, L.name.decode(), L.kh, L.cin, L.cout, t, tile, bad, a.size, n)
    det.debug_set_conv_tile(0)
    det.close()
    assert any(k.startswith("conv3x3_kx_kernel<128, 4, 2>") for k in names), names
    assert any(k.startswith("conv_igemm_kernel<128, 128, 4, 2, 3") for k in names), names
    assert n > 8 or any(k.startswith("conv_igemm_kernel<128, 192, 4, 2, 2") for k in names), names   # (n = 16: the halo kernel takes them)


def test_ring_give_up_word_is_an_error_not_wrong_results(rfd):
    """A ring wave that stops waiting marks the context's device fault word (the one the chunked NMS reports into); the host
    reads it back behind the pass and every entry point returns an error instead of tensors nobody can trust
    (the reference returns Err on every failure: face_detection.rs:498-509).  The word is poked from the host here -- a real
    give-up needs a wedged wave -- and must be cleared by the failing call."""
    import ctypes as C
    det = rfd.RetinaFaceDetection(image_size=(640, 640), max_batch_size=2, max_det=16)
    det.init_synthetic_weights(1234)
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    op = next(i for i in _ring_ops(g) if g.layers[g.ops[i].layer].kh == 3)
    det.debug_set_conv_tile(RING)
    det.debug_run(2, op, op)                                   # healthy
    L = rfd.load_library()
    assert L.rfd_debug_poke_nms_flag(det._ctx, 1) == 0
    with pytest.raises(rfd.RfdError) as e:
        det.debug_run(2, op, op)
    assert "gave up" in str(e.value)
    det.debug_run(2, op, op)                                   # the word was cleared: healthy again
    det.debug_set_conv_tile(0)
    det.close()

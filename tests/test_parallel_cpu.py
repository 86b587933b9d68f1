"""CPU tests of the N > 1 path: image-parallel sharding + the all-gather of the per-rank detection slabs
(rfd_hip.parallel), run with the gloo backend at world_size 2 (the same code runs over RCCL on GPU
tensors in bench.py).  Detections are produced by the CPU oracle here -- only the host logic
(sharding, slab layout, collective, unpacking) is under test; the HIP path needs a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dets_for(indices, H=128, W=128):
    from oracle import oracle as O
    out = []
    for i in indices:
        heads = [h[0] for h in helpers.make_heads(500 + i, 1, H, W, cand_rate=0.1, n_faces=3)]
        det, lmk, _, _ = O.decode_nms(heads, H, W, 0.7, 0.45, det_scale=0.5 + 0.01 * i)
        out.append((det, lmk))
    return out


def _worker(rank, world, port, total, max_det, q):
    for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from rfd_hip import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = parallel.shard_range(total, world, rank)
    per = -(-total // world)
    slab = parallel.DetectionSlab(per, max_det)          # fixed capacity per rank: ceil(total/world)
    slab.fill_from(_dets_for(range(lo, hi)))
    gathered, _ = parallel.gather_detections(slab)
    res = parallel.unpack_gathered(slab, gathered)
    # the layout of the C ABI's gather (rfd_gather_detections: four all-gathers, each array rank-major) with gloo
    # standing in for ncclAllGather: GatheredSlabs.unpack must give the same rows
    gs = parallel.GatheredSlabs(world, per, max_det, "cpu")
    nb, nl, n = gs.n_boxes, gs.n_lmk, world * per
    for dst, src in ((gs.buf[:nb], slab.buf[:slab.n_boxes]), (gs.buf[nb:nb + nl], slab.buf[slab.n_boxes:slab.n_boxes + slab.n_lmk]),
                     (gs.buf[nb + nl:nb + nl + n], slab.count()), (gs.buf[nb + nl + n:], slab.total())):
        dist.all_gather_into_tensor(dst, src.contiguous())
    res2, tot2 = gs.unpack()
    assert len(res2) == len(res) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(res, res2))
    if rank == 0:
        q.put([(d.tolist(), k.tolist()) for d, k in res])
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from rfd_hip import parallel
    for total in (1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) == -(-total // world)


def test_slab_roundtrip_and_truncation():
    from rfd_hip import parallel
    dets = _dets_for(range(3))
    slab = parallel.DetectionSlab(3, 8)
    slab.fill_from(dets)
    back = slab.unpack()
    for (d, k), (d2, k2) in zip(dets, back):
        n = min(len(d), 8)
        assert np.array_equal(d[:n], d2) and np.array_equal(k[:n], k2)
    assert slab.total().tolist() == [len(d) for d, _ in dets]
    pb, pl, pc, pt = slab.pointers()
    assert pl - pb == 3 * 8 * 5 * 4 and pc - pl == 3 * 8 * 10 * 4 and pt - pc == 3 * 4


@pytest.mark.parametrize("total", [6, 5])
def test_gather_world_size_2_matches_single_process(total):
    world, max_det = 2, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, max_det, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _dets_for(range(total))
    per = -(-total // world)
    assert len(res) == per * world                       # tail slots of the last rank are empty
    for i, (d, k) in enumerate(want):
        assert np.array_equal(np.array(res[i][0], np.float32).reshape(-1, 5), d[:max_det])
        assert np.array_equal(np.array(res[i][1], np.float32).reshape(-1, 5, 2), k[:max_det])
    for i in range(total, per * world):
        assert res[i][0] == []


def _sustain_worker(rank, world, port, q):
    import time
    import torch
    import torch.distributed as dist
    from rfd_hip import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.zeros(1)

    def step():                       # a step with a collective in it, and ranks of very different speed
        time.sleep(0.02 if rank == 0 else 0.001)
        dist.all_reduce(x)
    # rank 1 starts its window late: with per-rank stop decisions rank 0 would leave one chunk earlier and both would hang
    if rank == 1:
        time.sleep(0.15)
    el, steps = parallel.run_for_at_least(step, lambda: None, 0.5, 5, world, "cpu")
    dist.barrier()
    q.put((rank, steps, el))
    dist.destroy_process_group()


def test_sustained_window_stops_collectively_world_size_2():
    """bench.py's sustained-rate loop (round-3 advisor finding): all ranks must run the same number of steps although their
    clocks start apart and their steps differ in speed."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sustain_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and got[0][1] >= 5
    assert max(g[2] for g in got) >= 0.5


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` with WORLD_SIZE unset must start N rank processes itself (the driver's multi-GPU
    command), relay rank 0's single JSON line and return the children's status.  Rehearsed on CPU: RFD_BENCH_DRYRUN makes
    every rank stop after the rendezvous + one gloo all-reduce, before any GPU call."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["RFD_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d == {"dryrun": True, "n_gpus": 2, "rank_sum": 3, "steps": 3, "warmup": 1}
    # a failing rank makes the launcher fail
    env["RFD_BENCH_DRYRUN_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode != 0

"""Image-parallel sharding of a detection batch over one process per GPU, and the gather of the
per-rank detection slabs (RCCL all-gather over xGMI when the tensors live on the GPU; the same code
runs over gloo on CPU tensors in the tests).

The reference has no multi-device path at all (one image per call, face_detection.rs:220); images
are independent through every stage, so the only exchange step is this final gather.  Slab layout
per rank (int32 words, floats bit-cast):  boxes [B][max_det][5] f32 | landmarks [B][max_det][10] f32
| count [B] i32 | total [B] i32 -- the four arrays of `rfd_dets`, contiguous so that ONE collective
moves them (message ~ B*max_det*60 B: latency-bound over xGMI, not bandwidth-bound)."""
import numpy as np
import torch


def shard_range(total, world_size, rank):
    """Contiguous split of `total` images: ceil(total / world) per rank, the tail rank gets fewer."""
    per = -(-total // world_size)
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


class DetectionSlab:
    def __init__(self, batch, max_det, device="cpu"):
        self.batch, self.max_det = batch, max_det
        self.n_boxes = batch * max_det * 5
        self.n_lmk = batch * max_det * 10
        self.words = self.n_boxes + self.n_lmk + 2 * batch
        self.buf = torch.zeros(self.words, dtype=torch.int32, device=device)

    # views
    def boxes(self, buf=None):
        b = self.buf if buf is None else buf
        return b[:self.n_boxes].view(torch.float32).view(self.batch, self.max_det, 5)

    def landmarks(self, buf=None):
        b = self.buf if buf is None else buf
        return b[self.n_boxes:self.n_boxes + self.n_lmk].view(torch.float32).view(self.batch, self.max_det, 5, 2)

    def count(self, buf=None):
        b = self.buf if buf is None else buf
        o = self.n_boxes + self.n_lmk
        return b[o:o + self.batch]

    def total(self, buf=None):
        b = self.buf if buf is None else buf
        o = self.n_boxes + self.n_lmk + self.batch
        return b[o:o + self.batch]

    def pointers(self):
        """Device (or host) addresses of the four arrays, for rfd_dets."""
        base = self.buf.data_ptr()
        o = 4 * self.n_boxes
        return base, base + o, base + o + 4 * self.n_lmk, base + o + 4 * self.n_lmk + 4 * self.batch

    def fill_from(self, dets):
        """dets: list of (det [K,5], kps [K,5,2]) numpy pairs (used by the CPU tests)."""
        bx, lm, ct, tt = self.boxes(), self.landmarks(), self.count(), self.total()
        for i, (d, k) in enumerate(dets):
            n = min(len(d), self.max_det)
            bx[i, :n] = torch.from_numpy(np.ascontiguousarray(d[:n]))
            lm[i, :n] = torch.from_numpy(np.ascontiguousarray(k[:n]))
            ct[i] = n
            tt[i] = len(d)

    def unpack(self, buf=None):
        bx, lm, ct = self.boxes(buf).cpu().numpy(), self.landmarks(buf).cpu().numpy(), self.count(buf).cpu().numpy()
        return [(bx[i, :ct[i]].copy(), lm[i, :ct[i]].copy()) for i in range(self.batch)]


def gather_detections(slab, out=None, group=None, async_op=False):
    """All-gather every rank's slab: returns (gathered [world, words] int32 tensor, work handle)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty(world * slab.words, dtype=torch.int32, device=slab.buf.device)
    work = dist.all_gather_into_tensor(out, slab.buf, group=group, async_op=async_op)
    return out.view(world, slab.words), work


def unpack_gathered(slab, gathered):
    """[world, words] -> list over all images of the global batch (rank-major = original order)."""
    res = []
    for r in range(gathered.shape[0]):
        res += slab.unpack(gathered[r])
    return res


class GatheredSlabs:
    """Destination of rfd_gather_detections: `world` slabs of `batch` frames each, as four rank-major arrays
    (boxes [world*batch][max_det][5] | landmarks | count | total) in one device buffer."""

    def __init__(self, world, batch, max_det, device):
        self.world, self.batch, self.max_det = world, batch, max_det
        n = world * batch
        self.n_boxes, self.n_lmk = n * max_det * 5, n * max_det * 10
        self.buf = torch.zeros(self.n_boxes + self.n_lmk + 2 * n, dtype=torch.int32, device=device)

    def pointers(self):
        base = self.buf.data_ptr()
        o = 4 * self.n_boxes
        n = self.world * self.batch
        return base, base + o, base + o + 4 * self.n_lmk, base + o + 4 * self.n_lmk + 4 * n

    def unpack(self):
        n = self.world * self.batch
        b = self.buf.cpu()
        bx = b[:self.n_boxes].view(torch.float32).view(n, self.max_det, 5).numpy()
        lm = b[self.n_boxes:self.n_boxes + self.n_lmk].view(torch.float32).view(n, self.max_det, 5, 2).numpy()
        ct = b[self.n_boxes + self.n_lmk:self.n_boxes + self.n_lmk + n].numpy()
        tt = b[self.n_boxes + self.n_lmk + n:].numpy()
        return [(bx[i, :ct[i]].copy(), lm[i, :ct[i]].copy()) for i in range(n)], tt.copy()


def init_comm(det, group=None):
    """Build the detector's own RCCL communicator (C ABI: rfd_comm_init) for the ranks of a torch.distributed group:
    rank 0 draws the unique id, torch.distributed only carries its 128 bytes."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [det.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    det.comm_init(box[0], rank, world)
    return rank, world


def run_for_at_least(step, sync, seconds, chunk, world=1, device="cpu", clock=None):
    """Run `step()` in chunks of `chunk` calls, `sync()` behind every chunk, until `seconds` of wall time have passed on ANY
    rank; returns (elapsed seconds on this rank, steps run).  The decision to stop is collective (an all-reduce MAX of each
    rank's own verdict), so every rank runs the SAME number of steps: a step of a multi-rank job holds a collective (the RCCL
    all-gather of the detection slabs), and a rank that left the loop on its own clock while another enqueued one more chunk
    would hang both.  bench.py's sustained-rate window; world-size-2 test over gloo in tests/test_parallel_cpu.py."""
    import time
    import torch
    import torch.distributed as dist
    clock = clock or time.perf_counter
    t0 = clock()
    steps = 0
    while True:
        for _ in range(chunk):
            step()
        steps += chunk
        sync()
        t1 = clock()
        stop = torch.tensor([1.0 if t1 - t0 >= seconds else 0.0], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(stop, op=dist.ReduceOp.MAX)
        if float(stop.item()) > 0:
            return t1 - t0, steps

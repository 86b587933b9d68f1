#!/usr/bin/env python3
"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

metric : images/s (+ p50 latency) of the whole detection hot path -- letterbox preprocess, RetinaFace-R50
         forward (bf16 MFMA implicit-GEMM convs), decode, sort, NMS, rescale -- on 640x640 frames, batch 32
         per GPU (BASELINE.json configs[2]), frames already resident in HBM when the clock starts.
N > 1  : one process per GPU (torch.distributed, backend nccl = RCCL), every rank runs its own shard of 32
         frames (weak scaling) and the per-rank detection slabs are all-gathered over xGMI each step.
Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around every conv launch
(rfd_set_profiling); `cpu_baseline` times the restated CPU path (torch-CPU f32 forward + the C oracle) on a
bounded sample on rank 0 at N = 1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "rs-face-detection_amd", "python"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ASYNC_MODE = int(os.environ.get("RFD_BENCH_ASYNC", "2"))  # 1: every step ordered on one stream; 2: cross-step overlap
BATCH = int(os.environ.get("RFD_BENCH_BATCH", "32"))  # 32 is the headline configuration
IMAGE = 640
# source frame size (h, w); default = the network size.  RFD_BENCH_SRC=1080x1920 gives BASELINE.json configs[3]'s per-GPU slice
SRC_H, SRC_W = (int(v) for v in os.environ.get("RFD_BENCH_SRC", "640x640").split("x"))
MAX_DET = 1024
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
TARGET_CAND_RATE = 0.006   # ~100 candidates / image, a realistic operating point


def calibrate_cls_bias(det, graph, frames_np):
    """Random weights give arbitrary score statistics: shift the fg logits of the three heads so that
    TARGET_CAND_RATE of the anchors clear the 0.7 threshold (setup, outside the timed region)."""
    _, tensor, _ = det.preprocess(frames_np[:2])
    heads = det.forward(tensor)
    p = np.concatenate([heads[3 * l][:, 2:4].reshape(-1) for l in range(3)]).astype(np.float64)
    p = np.clip(p, 1e-7, 1 - 1e-7)
    d = np.log(p / (1 - p))
    delta = float(np.log(0.7 / 0.3) - np.quantile(d, 1.0 - TARGET_CAND_RATE))
    for i, L in enumerate(graph.layers):
        if L.name.decode().startswith("head"):
            w, b = det.get_layer(i, L)
            b[2:4] += delta
            det.set_layer(i, w, b)
    return delta


def cpu_baseline(det, graph, frames_np, thr):
    """Restated CPU path on the host cores: oracle preprocess -> torch-CPU f32 forward of the same graph and
    weights (stands in for the Triton-CPU backend) -> oracle decode + NMS.  Bounded sample."""
    import torch_ref
    from oracle import oracle as O
    ref = torch_ref.TorchRef(graph, det)
    n = 2
    t0 = time.perf_counter()
    reps = 0
    while True:
        pre = [O.preprocess(f, IMAGE, IMAGE) for f in frames_np[:n]]
        x = torch.from_numpy(np.stack([p[1] for p in pre]))
        x4 = torch.cat([x, torch.zeros(n, 1, IMAGE, IMAGE)], 1)
        heads = ref.heads(ref.forward(x4))
        for b in range(n):
            O.decode_nms([h[b] for h in heads], IMAGE, IMAGE, np.float32(thr), 0.45, float(pre[b][2]))
        reps += 1
        el = time.perf_counter() - t0
        if el > 12.0 or reps >= 8:
            break
    return {"value": round(n * reps / el, 3), "unit": "images/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d passes of %d frames 640x640: C oracle preprocess + torch-CPU f32 R50 forward "
                      "(same graph/weights, stands in for Triton-CPU) + C oracle decode/NMS" % (reps, n)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import helpers
    import rfd_hip
    from rfd_hip import parallel

    det = rfd_hip.RetinaFaceDetection(image_size=(IMAGE, IMAGE), max_batch_size=BATCH, device_id=local_rank,
                                      max_det=MAX_DET)
    det.init_synthetic_weights(1234)
    graph = rfd_hip.Graph(rfd_hip.BACKBONE_R50, IMAGE, IMAGE)

    # synthetic 640x640 BGR frames (seed 1 + global image index), resident in HBM
    frames_np = [helpers.make_image(1000 + rank * BATCH + i, SRC_H, SRC_W) for i in range(BATCH)]
    delta = calibrate_cls_bias(det, graph, frames_np)
    frames = torch.from_numpy(np.stack(frames_np)).to(dev)
    fptrs = [frames.data_ptr() + i * SRC_H * SRC_W * 3 for i in range(BATCH)]
    shapes = [(SRC_H, SRC_W)] * BATCH
    slab = parallel.DetectionSlab(BATCH, MAX_DET, device=dev)
    pb, pl, pc, pt = slab.pointers()
    gathered = torch.empty(world * slab.words, dtype=torch.int32, device=dev) if world > 1 else None

    stream = torch.cuda.current_stream()
    det.set_stream(stream.cuda_stream)  # detector, RCCL and torch share one stream order
    if os.environ.get("RFD_BENCH_TILE"):  # experiment knob: forced conv tile configuration (rfd_debug_set_conv_tile)
        det.debug_set_conv_tile(int(os.environ["RFD_BENCH_TILE"]))
    if os.environ.get("RFD_BENCH_CONC"):  # experiment knob: "multi_stream,split_min_part,split_max_parts,use_graph"
        ms, sp, mp_, gr = (int(v) for v in os.environ["RFD_BENCH_CONC"].split(","))
        det.debug_set_concurrency(bool(ms), sp, mp_, bool(gr))

    def step():
        # async 2: the frames are complete in HBM, so the chains of this step may overlap the previous step's tail
        det.detect_device(fptrs, shapes, pb, pl, pc, pt, async_=ASYNC_MODE)
        if world > 1:
            parallel.gather_detections(slab, out=gathered)

    # set-up pass, outside warm-up and timing: the network's one-off choice of chain streams (Network::tune_streams)
    # happens in the first split pass and takes about half a second
    det.detect_device(fptrs, shapes, pb, pl, pc, pt, async_=0)

    def fence():
        det.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    det.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if world > 1:
        dist.barrier()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # p50 latency of one synchronous step (batch of 32 end to end)
    lat = []
    for _ in range(min(args.steps, 20)):
        fence()
        a = time.perf_counter()
        step()
        det.sync()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    p50 = float(np.median(lat))

    counts = slab.count().cpu().numpy()
    totals = slab.total().cpu().numpy()

    # roofline of the dominant kernel class (implicit-GEMM convs): HIP events around every launch
    det.set_profiling(True)
    conv_ms, conv_flops, launches = [], 0.0, 0
    op_ms = np.zeros(graph.num_ops)
    reps = 5
    for _ in range(reps):
        det.detect_device(fptrs, shapes, pb, pl, pc, pt, async_=False)
        ms, fl, nl = det.conv_profile()
        conv_ms.append(ms)
        conv_flops, launches = fl, nl
        op_ms += det.op_profile(graph.num_ops)
    det.set_profiling(False)
    stats = det.stats()
    conv_ms_med = float(np.median(conv_ms))
    serial_tflops = conv_flops / (conv_ms_med * 1e-3) / 1e12
    op_ms /= reps
    # In the timed configuration the network runs as two overlapped half-batch chains (+ side streams), so single
    # launches cannot be timed in isolation; the class is timed as a whole: HIP events on the caller's stream at the
    # fork and after the join of the network pass (rfd_stats.ms_network), all conv FLOPs of the pass over that time.
    net_ms = []
    for _ in range(10):
        det.detect_device(fptrs, shapes, pb, pl, pc, pt, async_=False)
        net_ms.append(det.stats()["ms_network"])
    net_ms_med = float(np.median(net_ms))
    net_flops = 2.0 * graph.macs * BATCH
    achieved = net_flops / (net_ms_med * 1e-3) / 1e12

    # HBM traffic of the conv class: measured offline with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    # passes over this same command and corrected as MI355X_MICROARCH.md prescribes (tools/rocpd_summary.py);
    # it cannot be collected from inside the process, so the committed summary is reported (or null).
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
    if BATCH == 32 and os.path.exists(tpath):
        try:
            traffic = float(json.load(open(tpath))["conv_igemm_hbm_bytes_per_pass"])
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        value = world * BATCH * args.steps / elapsed
        out = {
            "metric": "images/sec + p50 latency, RetinaFace-R50 640x640 b32 (end-to-end detect: preprocess + CNN + decode + NMS)",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "RetinaFace-ResNet50 640x640 batch=32 bf16 per GPU (BASELINE.json configs[2])",
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world, "image_size": [IMAGE, IMAGE],
                       "source_frames": "%dx%dx3 u8 synthetic, resident in HBM" % (SRC_W, SRC_H), "max_det": MAX_DET,
                       "weights": "seeded synthetic (no model file exists in the reference), cls bias calibrated %+.3f" % delta,
                       "parallelism": "image-parallel x%d, RCCL all-gather of detection slabs" % world if world > 1 else "single GPU",
                       "candidates_per_image": round(float(stats["candidates"]) / BATCH, 1),
                       "detections_per_image": round(float(totals.mean()), 1)},
            "p50_ms": round(p50, 4),
            "p50_ms_per_image": round(p50 / BATCH, 5),
            "stage_ms": dict({k: round(stats[k], 4) for k in ("ms_preprocess", "ms_decode", "ms_sort", "ms_nms")},
                             ms_network=round(net_ms_med, 4)),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per forward pass of the network conv kernels (PMC, profiles/hbm_traffic_latest.json)",
                         "kernel": "network conv kernels of one forward pass (conv_igemm / conv3x3_kx / conv_b2b_s1 / stem), "
                                   "timed as a class: the two half-batch chains overlap",
                         "flops_per_pass": net_flops, "ms_per_pass": round(net_ms_med, 4),
                         "serialised": {"note": "same pass with every op on one stream, HIP events around each of the %d "
                                                "implicit-GEMM launches (conv0/stem excluded)" % launches,
                                        "flops_per_pass": conv_flops, "ms_per_pass": round(conv_ms_med, 4),
                                        "tflops": round(serial_tflops, 2),
                                        "avg_launch_us": round(conv_ms_med * 1e3 / max(launches, 1), 2)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(det, graph, frames_np, 0.7)
        else:
            out["cpu_baseline"] = None
        if os.environ.get("RFD_BENCH_OPS"):
            rows = []
            for i, o in enumerate(graph.ops):
                L = graph.layers[o.layer]
                t = graph.tensors[o.out if o.out >= 0 else (o.out2 if o.out2 >= 0 else o.outf)]
                fl = 2.0 * o.macs * BATCH
                rows.append("%3d %-22s k%d s%d %4d->%4d @%3dx%-3d %8.1f us %7.1f TF" % (
                    i, L.name.decode() if o.kind != 1 else "maxpool", L.kh, L.stride, L.cin, L.cout, t.height,
                    t.width, op_ms[i] * 1e3, fl / (op_ms[i] * 1e-3) / 1e12 if op_ms[i] > 0 else 0))
            sys.stderr.write("\n".join(rows) + "\n")
    det.set_stream(None)
    det.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

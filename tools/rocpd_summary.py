#!/usr/bin/env python3
"""Summarises rocprofv3 runs of bench.py (rocpd .db output of ROCm 7) into the files committed under profiles/.

  rocpd_summary.py stats  <kernel_trace.db>            -> kernel_stats.csv on stdout-like file + network busy time
  rocpd_summary.py pmc    <fetch.db> <write.db>        -> HBM bytes per forward pass of the network conv kernels

"pass" = one detect call = one preprocess_kernel launch.  Network class = conv_igemm / conv3x3_kx / conv_b2b_s1 /
stem kernels.  With the batch split the two parts' kernels overlap, so next to the per-kernel averages the summary
reports the UNION of the class's busy intervals per pass -- the figure bench.py's roofline uses (network wall time).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies wide reads / LDS-DMA at half their bytes -> x2;
WRITE_SIZE is exact; both in KiB."""
import json
import sqlite3
import statistics
import sys

NET = ("conv_igemm_kernel", "conv3x3_kx_kernel", "conv_b2b_s1_", "stem_kernel", "stem_persistent_kernel", "pw_stream_kernel",
       "conv3x3_c64_kernel", "conv3x3_halo_kernel", "pw_gemm_kernel", "pw_wide_kernel", "pw_b2b_kernel", "pw_pair_kernel",
       "conv_ring_kernel", "conv_tile256_kernel")   # every network kernel of csrc/kernels_conv.hip + kernels_ring.hip


def is_net(name):
    return any(t in name for t in NET)


def short(name):
    return name.replace("void ", "").split("(")[0]


def stats(path, out_csv, out_json):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    total = sum(v[1] for v in agg.values())
    with open(out_csv, "w") as f:
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%.1f,%.2f,%.2f\n' % (k, c, t, t / c, 100 * t / total))
    # Steady-state step = the window between two consecutive decode launches (one per detect call, in every execution
    # mode; the stream tuner's set-up passes launch no decode, so they fall outside).  Per window: union of the network
    # class's busy intervals (kernels of neighbouring calls overlap in the async = 2 mode, which is the point), their summed
    # durations and launch count.  Median over the last windows of the run = the timed loop.
    dec = [s for n, s, e in rows if "decode_kernel" in n]
    net = [(s, e) for n, s, e in rows if is_net(n)]
    busy, summed, counts, wall = [], [], [], []
    for a, b in list(zip(dec, dec[1:]))[-12:]:
        iv = sorted((max(s, a), min(e, b)) for s, e in net if e > a and s < b)
        if not iv:
            continue
        u, (cs, ce) = 0, iv[0]
        for s, e in iv[1:]:
            if s > ce:
                u += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        u += ce - cs
        busy.append(u / 1e6)
        summed.append(sum(e - s for s, e in iv) / 1e6)
        counts.append(sum(1 for s, e in net if a <= s < b))
        wall.append((b - a) / 1e6)
    out = {"windows": len(busy), "launches_per_step_median": statistics.median(counts) if counts else 0,
           "step_wall_ms_median": statistics.median(wall) if wall else None,
           "network_busy_ms_per_step_median": statistics.median(busy) if busy else None,
           "network_kernel_time_sum_ms_per_step_median": statistics.median(summed) if summed else None,
           "note": "step = window between consecutive decode launches (steady state of the timed loop); busy = union of the "
                   "start..end intervals of the network kernels inside it (the two chains and neighbouring calls overlap)"}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out))


def pmc(fetch_db, write_db, out_json):
    def load(path, counter):
        db = sqlite3.connect(path)
        per, calls = {}, {}
        for n, v in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            k = short(n)
            per[k] = per.get(k, 0.0) + v
            calls[k] = calls.get(k, 0) + 1
        return per, calls
    fetch, calls = load(fetch_db, "FETCH_SIZE")
    write, _ = load(write_db, "WRITE_SIZE")
    passes = max(calls.get("rfd::decode_kernel<false>", 0), 1)  # exactly one per pass in every execution mode
    rows, nf, nw = {}, 0.0, 0.0
    for k in sorted(set(fetch) | set(write)):
        f = 2.0 * fetch.get(k, 0.0) * 1024 / passes
        w = write.get(k, 0.0) * 1024 / passes
        rows[k] = {"calls_per_pass": round(calls.get(k, 0) / passes, 2), "fetch_bytes_per_pass": f, "write_bytes_per_pass": w}
        if is_net(k):
            nf += f
            nw += w
    out = {"passes": passes, "conv_igemm_hbm_bytes_per_pass": nf + nw, "network_fetch_bytes_per_pass": nf,
           "network_write_bytes_per_pass": nw, "class": list(NET),
           "correction": "FETCH_SIZE x2 (gfx950 wide-read tally), WRITE_SIZE x1, KiB -> bytes", "kernels": rows}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))


def mfma(db_path, out_json):
    """MFMA pipe utilisation per network kernel: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over all SIMDs) over the
    kernel's own cycles x 1024 SIMDs; kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)."""
    db = sqlite3.connect(db_path)
    acc = {}
    for n, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
        k = short(n)
        if not is_net(k):
            continue
        a = acc.setdefault(k, {})
        a[c] = a.get(c, 0.0) + v
        a["_n_" + c] = a.get("_n_" + c, 0) + 1
    rows, tb, tc = {}, 0.0, 0.0
    for k, a in acc.items():
        busy, gui = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0)
        cyc = gui / 8.0
        rows[k] = {"dispatches": a.get("_n_GRBM_GUI_ACTIVE", 0), "mfma_busy_cycles": busy, "kernel_cycles": cyc,
                   "mfma_util": busy / (cyc * 1024.0) if cyc else None}
        tb += busy
        tc += cyc
    out = {"network_mfma_util": tb / (tc * 1024.0) if tc else None,
           "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); kernels profiled one dispatch at a time",
           "kernels": rows}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps({"network_mfma_util": out["network_mfma_util"],
                      "per_kernel": {k: round(v["mfma_util"], 4) for k, v in rows.items() if v["mfma_util"] is not None}}))


def timeline(path, out_txt):
    """One pass of the timed configuration as a text timeline: every kernel with its queue, start and end relative to the
    pass's preprocess launch, plus how much of the pass had 0 / 1 / 2+ network kernels in flight."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = "select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")
    rows = db.execute(q).fetchall()
    starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[0]]
    with open(out_txt, "w") as f:
        f.write("columns of `kernels`: %s\n" % cols)
        if len(starts) < 6:
            f.write("too few passes\n")
            return
        # a pass in the middle of the run: from its first preprocess launch to the one two launches later (two chains)
        a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 2] if len(starts) // 2 + 2 < len(starts) else len(rows)
        t0 = rows[a][1]
        ev = []
        for r in rows[a:b]:
            f.write("%9.1f %9.1f %8.1f us  q=%s  %s\n" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "-", short(r[0])[:90]))
            if is_net(r[0]):
                ev.append((r[1], 1))
                ev.append((r[2], -1))
        ev.sort()
        lvl, last, hist = 0, None, {}
        for t, d in ev:
            if last is not None:
                hist[min(lvl, 3)] = hist.get(min(lvl, 3), 0) + (t - last)
            lvl += d
            last = t
        tot = sum(hist.values())
        f.write("network kernels in flight: " + ", ".join("%d: %.1f us (%.0f%%)" % (k, v / 1e3, 100.0 * v / tot) for k, v in sorted(hist.items())) + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])

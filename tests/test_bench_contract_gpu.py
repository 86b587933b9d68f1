"""bench.py prints ONE JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "images/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.02 < rf["frac"] < 1.0
    # traffic: PMC bytes per pass, measured by child rocprofv3 passes of this very run when the profiler is on the box (else the
    # committed summary, labelled); either way within a sane band of the algorithmic 13 GB of the graph as fused
    assert "traffic_source" in rf
    if rf["traffic"] is not None:
        assert 10e9 < rf["traffic"] < 25e9, rf["traffic"]
    assert rf["traffic_source"].startswith(("measured in this run", "committed summary", "none")), rf["traffic_source"]
    print("roofline.traffic = %s (%s)" % (rf["traffic"], rf["traffic_source"]))
    # the sustained window (>= 3 s of the same step) next to the short timed loop
    su = d["sustained"]
    assert su["seconds"] >= 3.0 and su["steps"] >= 10 and su["images_per_s"] > 0.8 * d["value"]
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    assert d["value"] > 100 * cb["value"]

#!/bin/bash
# Same-box A/B of bench.py configurations: each argument is "label|ENV=VAL ENV=VAL ..." (empty env = shipped default).
# Prints label, value (img/s), ms_per_step, sustained img/s, network ms per line; full JSON lines go to gpurun_out/ab_<tag>.jsonl
#   tools/ab_bench.sh tag "base|" "safe|RFD_HIP_LIB=tools/bin/librfd_hip_safewaits.so" ...
TAG="$1"; shift
mkdir -p gpurun_out
OUT="gpurun_out/ab_${TAG}.jsonl"
: > "$OUT"
for spec in "$@"; do
  label="${spec%%|*}"; envs="${spec#*|}"
  line=$(env $envs RFD_BENCH_HOST_PATH=0 RFD_BENCH_TRAFFIC=off RFD_BENCH_SUSTAIN_S=${SUSTAIN_S:-1.5} python bench.py --no-cpu-baseline --steps ${STEPS:-30} --warmup 5 2>>gpurun_out/ab_${TAG}.err | tail -1)
  echo "{\"label\": \"$label\", \"env\": \"$envs\", \"result\": $line}" >> "$OUT"
  python - "$label" "$line" <<'PY'
import json, sys
try:
    r = json.loads(sys.argv[2])
    print("%-28s %9.1f img/s  %7.3f ms/step  sustained %9.1f  net %7.3f ms  frac %.4f" % (sys.argv[1], r["value"], r["ms_per_step"], (r.get("sustained") or {}).get("images_per_s", 0), r["roofline"]["ms_per_pass"], r["roofline"]["frac"]))
except Exception as e:
    print("%-28s FAILED (%s): %s" % (sys.argv[1], e, sys.argv[2][:200]))
PY
done

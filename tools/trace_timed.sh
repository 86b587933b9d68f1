#!/bin/bash
# kernel trace of bench.py in the TIMED configuration (async 2, tuned chain streams) -> per-kernel stats + one pass's timeline
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-timed}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
export RFD_BENCH_HOST_PATH=0
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/stats -name "*_results.db" | head -1)
python3 $R/tools/rocpd_summary.py stats $DB $O/kernel_stats.csv $O/network_busy.json
python3 $R/tools/rocpd_summary.py timeline $DB $O/timeline.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null || true
grep -h '"metric"' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json || true
rm -rf $O/stats

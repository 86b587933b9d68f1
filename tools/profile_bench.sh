#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box; leaves only small summaries under gpurun_out/prof/ (copy the ones to
# be judged into profiles/).  (1) kernel trace of the TIMED configuration (async 2, tuned chain streams);
# (2) PMC counters in their own runs (no trace domains mixed in), with calls ordered on one stream and the stream tuner
# off: under PMC every dispatch runs alone anyway, and the tuner would add ~120 network passes without a decode launch.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export RFD_BENCH_HOST_PATH=0 RFD_BENCH_TRAFFIC=off RFD_BENCH_SUSTAIN=0
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/stats -name "*_results.db" | head -1)
python3 $R/tools/rocpd_summary.py stats $DB $O/kernel_stats.csv $O/network_busy.json
python3 $R/tools/rocpd_summary.py timeline $DB $O/timeline.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null || true
grep -h '"metric"' $O/bench_under_rocprof.log > $O/bench_under_rocprof.json || true
rm -rf $O/stats
export RFD_STREAM_TUNE=0 RFD_BENCH_ASYNC=1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/bench_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/bench_mfma.log 2>&1
python3 $R/tools/rocpd_summary.py mfma $(find $O/mfma -name "*_results.db" | head -1) $O/mfma_util.json
python3 $R/tools/rocpd_summary.py pmc $(find $O/fetch -name "*_results.db" | head -1) $(find $O/write -name "*_results.db" | head -1) $O/hbm_traffic.json
if [ -n "$RFD_PROFILE_NO_MFAST" ]; then
  RFD_NO_MFAST=1 rocprofv3 --pmc FETCH_SIZE -d $O/fetch2 -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/bench_fetch2.log 2>&1
  python3 $R/tools/rocpd_summary.py pmc $(find $O/fetch2 -name "*_results.db" | head -1) $(find $O/write -name "*_results.db" | head -1) $O/hbm_traffic_n_fastest.json
fi
rm -rf $O/fetch $O/fetch2 $O/write $O/mfma $O/bench_mfma.log $O/bench_fetch.log $O/bench_fetch2.log $O/bench_write.log $O/bench_under_rocprof.log

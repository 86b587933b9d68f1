#!/bin/bash
# PMC counters of single network ops (tools/op_bench.py under rocprofv3 --pmc, one counter group per pass; no trace domains mixed in).
# usage: tools/pmc_ops.sh <ops: 20,11> <out.txt>     (on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
OPS=$1; OUT=$R/${2:-gpurun_out/pmc_ops.txt}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_ACTIVE_INST_ANY"; do
  D=/tmp/pmc_$$; rm -rf $D
  rocprofv3 --pmc $grp -d $D --output-format csv -- python3 $R/tools/op_bench.py --ops $OPS --tiles 0 --batches 16 --reps 3 --rounds 1 > /tmp/pmc_log_$$.txt 2>&1 || { echo "group [$grp] failed" >> $OUT; tail -3 /tmp/pmc_log_$$.txt >> $OUT; continue; }
  F=$(find $D -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$grp" >> $OUT <<'PY'
import csv, sys, collections
f, grp = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-70:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    if "rfd::" not in k: continue
    print("%-72s %s" % (k, "  ".join("%s=%.4g" % (c, acc[k][c] / max(n[(k, c)], 1)) for c in grp.split() if c in acc[k])))
PY
done
cat $OUT

#!/bin/bash
# usage: tools/pmc_probe.sh <outfile> <probe args...>   — PMC passes over tools/gemm3_probe.py (GPU box only)
out=$1; shift
cd /tmp; export TMPDIR=/tmp
: > $GRAFT_REPO_ROOT/$out
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "TCC_HIT TCC_MISS TCC_REQ" "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  d=$(mktemp -d /tmp/pmcXXXX)
  rocprofv3 --pmc $set --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/gemm3_probe.py "$@" > /dev/null 2>&1 || { echo "pass failed: $set" >> $GRAFT_REPO_ROOT/$out; continue; }
  python3 - "$d" >> $GRAFT_REPO_ROOT/$out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16x3" not in r["Kernel_Name"]:
            continue
        a = agg[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, (n, t) in sorted(agg.items()):
    print(f"{k:32s} launches {n:4d}  avg/launch {t / n:16.1f}")
PY
done

#!/bin/bash
# one GPU-box call of the edit-measure loop: quick parity subset, K sweep of the split GEMMs, one bench line per mode
set -o pipefail
OUT=gpurun_out/${1:-try}
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] parity subset"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "${2:-tile_configurations or gemm_bf16x3 or gemm_f16x2 or core_bf16x3 or full_step_f16x2 or full_width}" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
echo "[$(date +%T)] ksweep"
timeout -k 10 300 python3 tools/micro/s3_ksweep.py --iters 10 > $OUT/ksweep.txt 2>&1 || { tail $OUT/ksweep.txt; exit 1; }
AVD_S3_TILE=1 timeout -k 10 300 python3 tools/micro/s3_ksweep.py --iters 10 --only res > $OUT/ksweep_tile1.txt 2>&1 || { tail $OUT/ksweep_tile1.txt; exit 1; }
AVD_S3_TILE=0 timeout -k 10 300 python3 tools/micro/s3_ksweep.py --iters 10 --only gelu_split > $OUT/ksweep_tile0.txt 2>&1 || { tail $OUT/ksweep_tile0.txt; exit 1; }
grep -h "N=" $OUT/ksweep.txt $OUT/ksweep_tile1.txt $OUT/ksweep_tile0.txt
echo "[$(date +%T)] bench"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", round(d["value"],1), d["config"]["matmul"], "err", d.get("parity_rel_err_vs_cpu_oracle"), "roof", d["roofline"]["kernel"], round(d["roofline"]["frac"],3), "cpu", d.get("cpu_baseline",{}).get("value"))
for a in d.get("alt",[])+[d.get("speed_mode",{})]:
    if a: print(" alt", a["matmul"], round(a["value"],1), a.get("parity_rel_err_vs_cpu_oracle"), a.get("roofline",{}).get("frac"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["ms_per_step"])[:8]:
    print("  %-50s %.3f ms/step x%.0f" % (k[:50], v["ms_per_step"], v["launches_per_step"]))
PY
echo "[$(date +%T)] done"

#!/bin/bash
# round-4 GPU call K: where does the split-operand path start to win now that it has short blocks?  s3_min_rows 6144 (default) vs 1
set -o pipefail
OUT=gpurun_out/r4k
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; for mr in 6144 1; do echo "[$(date +%T)] $name min_rows=$mr: $*"; AVD_S3_MIN_ROWS=$mr timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt "$@" > $OUT/bench_${name}_mr$mr.json 2>>$OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_${name}_mr$mr.json > $OUT/k.txt; head -7 $OUT/k.txt; done; }
run 128_b16 --size 128 --batch 16
run 128_b8 --size 128 --batch 8
run c2 --size 64 --batch 32
run c2_b16 --size 64 --batch 16
run c3_b4 --batch 4
run c3_b2 --batch 2
echo "[$(date +%T)] done"

#!/bin/bash
# round-4 GPU call M: whole GPU tier with the per-mode row threshold, then the small / mid configurations at their defaults
set -o pipefail
OUT=gpurun_out/r4m
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest -m gpu"
timeout -k 10 1000 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1 || { grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20; tail -5 $OUT/pytest.log; echo "pytest failed: no bench lines from a failing tree"; exit 1; }
tail -2 $OUT/pytest.log
run() { name=$1; shift; echo "[$(date +%T)] $name: $*"; timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt "$@" > $OUT/bench_$name.json 2>>$OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_$name.json > $OUT/k.txt; head -4 $OUT/k.txt; }
run c1 --size 32 --batch 4
run c2 --size 64 --batch 32
run c2_graph --size 64 --batch 32 --graph
run 128_b8 --size 128 --batch 8
run 128_b16 --size 128 --batch 16
run 128 --size 128
run c3_b2 --batch 2
run c3_b4 --batch 4
run c3_b8 --batch 8
run c3
echo "[$(date +%T)] done"

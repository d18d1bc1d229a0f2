#!/bin/bash
# round-4 GPU call J: whole GPU tier after the short-block / deep-ring / register-epilogue routing, then the configurations' step rates
set -o pipefail
OUT=gpurun_out/r4j
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest -m gpu"
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1 || { grep -E "FAILED|Error" $OUT/pytest.log | head; tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
run() { name=$1; shift; echo "[$(date +%T)] $name: $*"; timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt "$@" > $OUT/bench_$name.json 2>>$OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_$name.json > $OUT/k.txt; head -12 $OUT/k.txt; }
run 128 --size 128
run c3
run 128_b16 --size 128 --batch 16
run c3_b8 --batch 8
run c3_b16 --batch 16
run c5 --size 512 --batch 8
echo "[$(date +%T)] done"

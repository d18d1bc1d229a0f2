#!/bin/bash
# round-4 GPU call N: four-stage ring for in_proj / fc1 launches that fit the CUs once — parity subset, A/B at the small shapes
set -o pipefail
OUT=gpurun_out/r4n
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "short_four_wave or two_stream or full_step" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
run() { name=$1; shift; for deep in 0 1; do echo "[$(date +%T)] $name deep4=$deep: $*"; AVD_S3_DEEP4=$deep timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt "$@" > $OUT/bench_${name}_d$deep.json 2>>$OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_${name}_d$deep.json > $OUT/k.txt; head -5 $OUT/k.txt; done; }
run c2 --size 64 --batch 32
run 128_b8 --size 128 --batch 8
run c3_b4 --batch 4
run 128_b16 --size 128 --batch 16
echo "[$(date +%T)] done"

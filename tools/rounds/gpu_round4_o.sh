#!/bin/bash
# round-4 GPU call O: noise-head launches on short blocks / the four-stage ring (register bias epilogue) — parity, step rates, and where the head's split path starts to pay
set -o pipefail
OUT=gpurun_out/r4o
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest"
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "short_four_wave or two_stream or full_step or head or golden or default_mode" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
run() { name=$1; env=$2; shift; shift; echo "[$(date +%T)] $name $env: $*"; env $env timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt "$@" > $OUT/bench_$name.json 2>>$OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_$name.json > $OUT/k_$name.txt; head -1 $OUT/k_$name.txt; grep -E "kernel<(0|5|8)," $OUT/k_$name.txt; }
run 128_deep0 AVD_S3_DEEP4=0 --size 128
run 128 A=1 --size 128
run 128_b16 A=1 --size 128 --batch 16
run 128_b16_mr1 AVD_S3_MIN_ROWS=1 --size 128 --batch 16
run 128_b8 A=1 --size 128 --batch 8
run 128_b8_mr1 AVD_S3_MIN_ROWS=1 --size 128 --batch 8
run c2 A=1 --size 64 --batch 32
run c2_mr1 AVD_S3_MIN_ROWS=1 --size 64 --batch 32
run c3 A=1
echo "[$(date +%T)] done"

#!/bin/bash
# round-4 GPU call L: 64..128-row residual blocks — parity, then small shapes on the split path per forced RT (s3_min_rows 1)
set -o pipefail
OUT=gpurun_out/r4l
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "short_four_wave or two_stream" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
run() { name=$1; shift; for rt in 5 4 3 2 0; do echo "[$(date +%T)] $name rt4=$rt: $*"; AVD_S3_RT4=$rt AVD_S3_MIN_ROWS=1 timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt "$@" > $OUT/bench_${name}_rt$rt.json 2>>$OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_${name}_rt$rt.json > $OUT/k.txt; head -3 $OUT/k.txt; done; }
run c2 --size 64 --batch 32
run 128_b8 --size 128 --batch 8
run c3_b2 --batch 2
run 128_b16 --size 128 --batch 16
echo "[$(date +%T)] done"

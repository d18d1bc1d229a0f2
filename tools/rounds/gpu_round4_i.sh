#!/bin/bash
# round-4 GPU call I: short 4-wave blocks (s3_rt4) — parity, then the shipped 128x128 geometry and C3 per forced RT and automatic
set -o pipefail
OUT=gpurun_out/r4i
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "short_four_wave or block_rows or two_stream or full_step_shipped" -s > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
grep -E "short 4-wave|passed|failed" $OUT/pytest.log
for rt in 8 5 0; do for deep in 0 1; do
  echo "[$(date +%T)] 128x128 B=32 s3_rt4=$rt deep4=$deep"
  AVD_S3_DEEP4=$deep AVD_S3_RT4=$rt timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --size 128 --no-cpu-baseline --no-alt > $OUT/bench_128_rt${rt}_d$deep.json 2>>$OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_128_rt${rt}_d$deep.json > $OUT/k.txt; head -8 $OUT/k.txt
done; done
for rt in 8 0; do
  echo "[$(date +%T)] C3 s3_rt4=$rt"
  AVD_S3_RT4=$rt timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt > $OUT/bench_c3_rt$rt.json 2>>$OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_c3_rt$rt.json > $OUT/k.txt; head -7 $OUT/k.txt
done
echo "[$(date +%T)] done"

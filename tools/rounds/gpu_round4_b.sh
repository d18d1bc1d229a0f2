#!/bin/bash
# round-4 GPU call B: attention parity subset with the pipelined kernel, then bench lines with attn_pipe 0 / 1 (interleaved twice)
set -o pipefail
OUT=gpurun_out/r4b
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest attention + steps"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "attention or full_step or chain or core_bf16x3 or core_f16x2 or two_stream or class_default or shipped" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
for rep in 1 2; do
for p in 0 1; do
  echo "[$(date +%T)] bench attn_pipe=$p rep $rep"
  AVD_ATTN_PIPE=$p timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline > $OUT/bench_p${p}_$rep.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_p${p}_$rep.json | head -8
done
done
AVD_ATTN_PIPE=1 timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline --matmul f16x2 > $OUT/bench_f16x2_p1.json 2>> $OUT/bench.err || exit 1
AVD_ATTN_PIPE=0 timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline --matmul f16x2 > $OUT/bench_f16x2_p0.json 2>> $OUT/bench.err || exit 1
python3 tools/bench_kernels.py $OUT/bench_f16x2_p1.json | head -7
python3 tools/bench_kernels.py $OUT/bench_f16x2_p0.json | head -7
AVD_ATTN_PIPE=1 timeout -k 10 300 python3 bench.py --size 512 --batch 8 --steps 20 --warmup 3 --no-alt --no-cpu-baseline > $OUT/bench_c5_p1.json 2>> $OUT/bench.err || exit 1
AVD_ATTN_PIPE=0 timeout -k 10 300 python3 bench.py --size 512 --batch 8 --steps 20 --warmup 3 --no-alt --no-cpu-baseline > $OUT/bench_c5_p0.json 2>> $OUT/bench.err || exit 1
python3 tools/bench_kernels.py $OUT/bench_c5_p1.json | head -5
python3 tools/bench_kernels.py $OUT/bench_c5_p0.json | head -5
python3 -c "
import json
d=json.load(open('$OUT/bench_p1_2.json')); print('power_clock', d.get('power_clock'))"
echo "[$(date +%T)] done"

#!/bin/bash
# round-4 GPU call G: fp32 split-K test, C1 / C2 lines with and without it, default bench
set -o pipefail
OUT=gpurun_out/r4g
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -q -s -k "f32_splitk or full_step_vs_oracle or chained" > $OUT/pytest.log 2>&1 || { grep -E "FAILED|assert" $OUT/pytest.log | head; exit 1; }
tail -2 $OUT/pytest.log; grep "fp32 split-K" $OUT/pytest.log
B="timeout -k 10 300 python3 bench.py --no-alt --no-cpu-baseline"
for p in 0 4; do
  AVD_GEMM_SPLITK=$p $B --size 32 --batch 4 --steps 300 --warmup 30 --matmul f32 > $OUT/bench_c1_sk$p.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
  echo "C1 gemm_splitk=$p:"; python3 tools/bench_kernels.py $OUT/bench_c1_sk$p.json > $OUT/k.txt; head -8 $OUT/k.txt
done
$B --size 32 --batch 4 --steps 300 --warmup 30 > $OUT/bench_c1_auto.json 2>> $OUT/bench.err || exit 1
echo "C1 auto: $(python3 tools/bench_kernels.py $OUT/bench_c1_auto.json | head -1)"
$B --size 32 --batch 4 --steps 300 --warmup 30 --graph > $OUT/bench_c1_graph.json 2>> $OUT/bench.err || exit 1
echo "C1 auto graph: $(python3 tools/bench_kernels.py $OUT/bench_c1_graph.json | head -1)"
$B --size 64 --batch 32 --steps 100 --warmup 10 --matmul f32 > $OUT/bench_c2_f32.json 2>> $OUT/bench.err || exit 1
echo "C2 f32: $(python3 tools/bench_kernels.py $OUT/bench_c2_f32.json | head -1)"
$B --size 64 --batch 32 --steps 100 --warmup 10 --matmul bf16 > $OUT/bench_c2_bf16.json 2>> $OUT/bench.err || exit 1
echo "C2 bf16: $(python3 tools/bench_kernels.py $OUT/bench_c2_bf16.json | head -1)"
echo "[$(date +%T)] bench default"
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2>> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 -c "
import json
d=json.load(open('$OUT/bench.json')); print('value', d['value'], 'parity', d['parity_rel_err_vs_cpu_oracle'], 'roof', d['roofline']['kernel'], round(d['roofline']['frac'],3), 'cpu', d['cpu_baseline']['value']); print('power_clock', d.get('power_clock')); print('speed', d['speed_mode']['value'], [(a['matmul'], round(a['value'],1)) for a in d['alt']])"

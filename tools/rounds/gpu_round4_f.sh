#!/bin/bash
# round-4 GPU call F: the whole GPU test tier, then the default bench line
set -o pipefail
OUT=gpurun_out/r4f
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest -m gpu"
timeout -k 10 1100 python3 -m pytest tests -m gpu -q -s -k "two_stream or fused_mlp or splitk or f32_splitk or full_step_vs_oracle or chained or last_block" > $OUT/pytest.log 2>&1 || { grep -E "FAILED|Error|assert" $OUT/pytest.log | head -30; tail -5 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
grep -E "fused MLP" $OUT/pytest.log | head
echo "[$(date +%T)] bench"
timeout -k 10 500 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 tools/bench_kernels.py $OUT/bench.json > $OUT/k.txt; head -12 $OUT/k.txt
python3 -c "
import json
d=json.load(open('$OUT/bench.json')); print('value', d['value'], 'parity', d['parity_rel_err_vs_cpu_oracle'], 'roof', d['roofline']['kernel'], round(d['roofline']['frac'],3), 'cpu', d['cpu_baseline']['value']); print('power_clock', d.get('power_clock')); print('speed', d['speed_mode']['value'], [(a['matmul'], round(a['value'],1)) for a in d['alt']])"
echo "[$(date +%T)] done"

#!/bin/bash
# round-4 GPU call A: full GPU test tier (prints kept), LDS half-read calibration, baseline bench line
set -o pipefail
OUT=gpurun_out/r4a
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] micro lds_half"
true
true
echo "[$(date +%T)] pytest -m gpu"
timeout -k 10 1000 python3 -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
grep -E "rel err|d_model 1024|128x128 batch|fp8 attention|C5 step" $OUT/pytest.log | head -40
echo "[$(date +%T)] bench"
timeout -k 10 500 python3 bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 tools/bench_kernels.py $OUT/bench.json 2>/dev/null | head -30
echo "[$(date +%T)] done"

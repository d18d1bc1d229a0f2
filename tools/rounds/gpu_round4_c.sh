#!/bin/bash
# round-4 GPU call C: trimmed last block — parity subset, then bench with core_trim 0 / 1
set -o pipefail
OUT=gpurun_out/r4c
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "last_block or full_step or chain or golden or core_bf16x3 or engine or two_stream or default_mode or stale" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
for rep in 1 2; do
for p in 0 1; do
  echo "[$(date +%T)] bench core_trim=$p rep $rep"
  AVD_CORE_TRIM=$p timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline > $OUT/bench_t${p}_$rep.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
  python3 tools/bench_kernels.py $OUT/bench_t${p}_$rep.json > $OUT/k.txt; head -3 $OUT/k.txt
done
done
echo "[$(date +%T)] done"

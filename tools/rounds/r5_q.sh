#!/bin/bash
# round-5 GPU call Q: one-block-per-CU launches of the four-wave split GEMM (four-stage ring) with the fragments of tile kt + 1 read during step kt
# (default build, S3_PF) against all reads at the top of the step (-DS3_PF=0):
# parity subset, then the mid-size bench lines with each library, alternating.
set -o pipefail
OUT=gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$(pwd)/tools/micro/libavdiff_nopf.so
echo "[$(date +%T)] parity subset (default build)"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_f16x2.py -m gpu -x -q -k "split_gemm or full_step or head_split or default_mode or shipped or chain" > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.txt
B="timeout -k 10 300 python3 bench.py --no-alt --no-cpu-baseline --steps 100 --warmup 10"
for r in 1 2; do
  for cfg in "128_b32:--size 128 --batch 32" "128_b16:--size 128 --batch 16" "128_b8:--size 128 --batch 8" "c2:--size 64 --batch 32 --sampler-steps 100" "c3_b8:--batch 8"; do
    name=${cfg%%:*}; args=${cfg#*:}
    $B $args > $OUT/pf_${name}_$r.json 2>> $OUT/bench.err
    AVDIFF_HIP_LIB=$OLD $B $args > $OUT/nopf_${name}_$r.json 2>> $OUT/bench.err
  done
done
python3 - <<'PY'
import json,glob,collections
res=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/r5q/*_*.json')):
    try:
        d=json.load(open(f)); k=f.split('/')[-1].rsplit('_',1)[0]; res[k].append(round(d['value'],1))
    except Exception as e: print(f,'ERR',e)
for k in sorted(res): print(k, res[k])
PY
echo "[$(date +%T)] done"

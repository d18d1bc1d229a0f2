#!/bin/bash
# round-5 GPU call J: [h|m] A fragments of the four-wave split GEMM as a half-wave re-read into the [h|l] registers (default build) against
# the full re-read (tools/micro/libavdiff_fullread.so, -DAVD_S3_FULLREAD): parity subset, per-launch clock / power, bench A/B.
set -o pipefail
OUT=gpurun_out/r5j
mkdir -p $OUT
export TMPDIR=/tmp
FULL=$(pwd)/tools/micro/libavdiff_fullread.so
echo "[$(date +%T)] parity subset (default build)"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_f16x2.py -m gpu -x -q -k "split_gemm or full_step_c3 or bf16x3 or head_split or default_mode" > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.txt
for r in 1 2; do
  echo "[$(date +%T)] kernel power round $r"
  timeout -k 10 200 python3 tools/micro/kernel_power.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kp_half.txt
  AVDIFF_HIP_LIB=$FULL timeout -k 10 200 python3 tools/micro/kernel_power.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/kp_full.txt
done
AB="timeout -k 10 300 python3 bench.py --no-alt --no-cpu-baseline --steps 40 --warmup 5"
for r in 1 2; do
  echo "[$(date +%T)] bench round $r"
  $AB > $OUT/bench_half_$r.json 2>> $OUT/bench.err
  AVDIFF_HIP_LIB=$FULL $AB > $OUT/bench_full_$r.json 2>> $OUT/bench.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5j/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],2), d.get('parity_rel_err_vs_cpu_oracle'))
    except Exception as e: print(f, 'ERR', e)
PY
echo "[$(date +%T)] done"

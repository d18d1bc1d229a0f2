#!/bin/bash
# Round 5, GPU call H: the energy floor of the six-term MFMA stream; the default bench line re-taken with the round's PMC traffic file in place
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5h; mkdir -p $OUT; export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*" | tee -a $OUT/progress.log; }
step "mfma energy"
timeout -k 10 120 tools/micro/mfma_energy > $OUT/mfma_energy.txt 2>&1 || step "mfma_energy failed"
cat $OUT/mfma_energy.txt
step "kernel power (same box)"
timeout -k 10 300 python3 tools/micro/kernel_power.py 2>&1 | grep -v amdgpu > $OUT/kernel_power.txt || step "kernel_power failed"
cat $OUT/kernel_power.txt
step "bench"
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || step "bench failed"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5h/bench.json') if l.startswith('{')][-1])
print(d['value'], d['roofline'], d.get('power_clock'), d['cpu_baseline'])
PY
step done

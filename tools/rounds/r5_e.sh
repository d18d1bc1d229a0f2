#!/bin/bash
# Round 5, GPU call E: 16x16x32 attention (tests, A/B), the row-form CFG kernel, per-kernel power / clock
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5e; mkdir -p $OUT; export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*" | tee -a $OUT/progress.log; }
step "pytest attention + cfg"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "attention or cfg_unpatch or golden or full_step_vs_oracle or fp8" > $OUT/pytest_sel.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.log
tail -15 $OUT/pytest_sel.txt
step "kernel power"
timeout -k 10 300 python3 tools/micro/kernel_power.py > $OUT/kernel_power.txt 2>&1 || step "kernel_power failed"
AVD_ATTN_M16=1 timeout -k 10 300 python3 tools/micro/kernel_power.py > $OUT/kernel_power_m16.txt 2>&1 || step "kernel_power m16 failed"
B="timeout -k 10 300 python3 bench.py --no-alt --no-cpu-baseline --steps 40 --warmup 5"
step "bench"
$B > $OUT/bench_default.json 2> $OUT/bench.err || step "bench failed"
AVD_ATTN_M16=1 $B > $OUT/bench_m16.json 2>> $OUT/bench.err || step "bench failed"
$B > $OUT/bench_default2.json 2>> $OUT/bench.err || step "bench failed"
AVD_ATTN_M16=1 $B > $OUT/bench_m16_2.json 2>> $OUT/bench.err || step "bench failed"
AVD_CFG_ROWS=0 $B > $OUT/bench_cfg_gather.json 2>> $OUT/bench.err || step "bench failed"
$B --size 512 --batch 8 --steps 20 --warmup 3 > $OUT/bench_c5.json 2>> $OUT/bench.err || step "bench failed"
AVD_ATTN_M16=1 $B --size 512 --batch 8 --steps 20 --warmup 3 > $OUT/bench_c5_m16.json 2>> $OUT/bench.err || step "bench failed"
step done
grep -v amdgpu $OUT/kernel_power.txt; grep -v amdgpu $OUT/kernel_power_m16.txt | tail -1
for f in $OUT/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    k=d.get('kernels',{})
    at=[v['ms_per_step']/v['launches_per_step']*1e3 for n,v in k.items() if n.startswith('attn_')]
    cf=[(v['ms_per_step']*1e3, v.get('gbs')) for n,v in k.items() if n.startswith('cfg_')]
    print(sys.argv[1].split('/')[-1], round(d['value'],2), 'attn us', [round(a,1) for a in at], 'cfg', cf, 'parity', d.get('parity_rel_err_vs_cpu_oracle'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done

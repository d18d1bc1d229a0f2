#!/bin/bash
# Round 5, GPU call D: peeled attention pipeline + 16x4 super-tiles: GPU test tier, then bench lines (default, the plain attention kernel, stagger off)
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5d; mkdir -p $OUT; export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*" | tee -a $OUT/progress.log; }
step "pytest gpu"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.log
tail -15 $OUT/pytest_gpu.txt
B="timeout -k 10 300 python3 bench.py --no-alt --no-cpu-baseline --steps 40 --warmup 5"
step "bench"
$B > $OUT/bench_default.json 2> $OUT/bench.err || step "bench failed"
AVD_ATTN_PIPE=0 $B > $OUT/bench_attn_plain.json 2>> $OUT/bench.err || step "bench failed"
AVD_S3_STAGGER=0 $B > $OUT/bench_stagger0.json 2>> $OUT/bench.err || step "bench failed"
AVD_S3_SN=16 AVD_S3_SUPER4=32 $B > $OUT/bench_old_supertile.json 2>> $OUT/bench.err || step "bench failed"
$B > $OUT/bench_default2.json 2>> $OUT/bench.err || step "bench failed"
$B --size 512 --batch 8 --steps 20 --warmup 3 > $OUT/bench_c5.json 2>> $OUT/bench.err || step "bench failed"
$B --matmul f16x2 > $OUT/bench_f16x2.json 2>> $OUT/bench.err || step "bench failed"
AVD_S3_SN=16 AVD_S3_SUPER4=32 $B --matmul f16x2 > $OUT/bench_f16x2_old_supertile.json 2>> $OUT/bench.err || step "bench failed"
$B --size 128 --batch 32 --steps 100 --warmup 10 > $OUT/bench_128.json 2>> $OUT/bench.err || step "bench failed"
step done
for f in $OUT/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    k=d.get('kernels',{})
    at=[v['ms_per_step']/v['launches_per_step']*1e3 for n,v in k.items() if n.startswith('attn_')]
    print(sys.argv[1].split('/')[-1], round(d['value'],2), 'attn us', [round(a,1) for a in at], 'parity', d.get('parity_rel_err_vs_cpu_oracle'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done

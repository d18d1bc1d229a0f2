#!/bin/bash
# Everything the round's measurement record is built from, in one GPU-box call (run from the repo root):
#   bash tools/profile_round.sh r02
# Output under gpurun_out/<tag>/; the summaries that are judged get copied into profiles/ (see the end of this script).
set -o pipefail
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*"; }

step "bench lines"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
python3 bench.py --matmul f32 --no-alt > $OUT/bench_f32.json 2>> $OUT/bench.err || exit 1
python3 bench.py --size 64 --batch 32 --sampler-steps 100 --steps 100 --warmup 10 --matmul bf16 --no-alt > $OUT/bench_c2_bf16.json 2>> $OUT/bench.err || exit 1
python3 bench.py --size 64 --batch 32 --sampler-steps 100 --steps 100 --warmup 10 --matmul f32 --no-alt > $OUT/bench_c2_f32.json 2>> $OUT/bench.err || exit 1
python3 bench.py --size 512 --batch 8 --steps 20 --warmup 3 --attn fp8 --no-alt --cpu-steps 1 > $OUT/bench_c5_fp8attn.json 2>> $OUT/bench.err || exit 1
python3 bench.py --size 512 --batch 8 --steps 20 --warmup 3 --no-alt --no-cpu-baseline > $OUT/bench_c5.json 2>> $OUT/bench.err || exit 1
python3 bench.py --size 32 --batch 4 --steps 200 --warmup 20 --no-alt --cpu-steps 3 --matmul f32 > $OUT/bench_c1_gpu.json 2>> $OUT/bench.err || exit 1

step "rocprofv3 kernel traces"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bf16x3 -o run -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $OUT/prof_bf16x3.log 2>&1) || exit 1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f32 -o run -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --matmul f32 > $OUT/prof_f32.log 2>&1) || exit 1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_vae -o run -- python3 $ROOT/tools/vae_bench.py --iters 5 > $OUT/prof_vae.log 2>&1) || exit 1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_vae3 -o run -- python3 $ROOT/tools/vae_bench.py --iters 5 --matmul bf16x3 > $OUT/prof_vae3.log 2>&1) || exit 1

step "PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes)"
python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic_bf16x3.json > $OUT/traffic_bf16x3.txt 2>&1 || exit 1
python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic.json -- --matmul f32 > $OUT/traffic_f32.txt 2>&1 || exit 1

step "PMC utilisation"
python3 tools/pmc_util.py --out gpurun_out/$TAG/util_bf16x3.json > $OUT/util_bf16x3.txt 2>&1 || exit 1
python3 tools/pmc_util.py --out gpurun_out/$TAG/util_f32.json -- --matmul f32 > $OUT/util_f32.txt 2>&1 || exit 1

step "N=2 rehearsal (gloo, both ranks on this one GPU)"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 \
    --backend gloo --share-device --no-alt > $OUT/bench_n2.out 2> $OUT/bench_n2.err || exit 1
grep '^{' $OUT/bench_n2.out > $OUT/bench_n2_gloo_rehearsal.json || exit 1

step "end to end + soak"
python3 tools/e2e_bench.py > $OUT/e2e.txt 2>&1 || echo "e2e failed"
step done

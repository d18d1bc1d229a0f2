#!/bin/bash
# Everything the round's measurement record is built from, in one GPU-box call (run from the repo root):
#   bash tools/profile_round.sh r02
# Output under gpurun_out/<tag>/; tools/collect_profiles.py copies the judged summaries into profiles/.
set -o pipefail
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*"; }
B="python3 bench.py"

step "bench lines"
$B > $OUT/bench.json 2> $OUT/bench.err || exit 1                                                     # default: f16x2, C3
$B --matmul f32 --no-alt > $OUT/bench_f32.json 2>> $OUT/bench.err || exit 1
$B --matmul bf16x3 --no-alt > $OUT/bench_bf16x3.json 2>> $OUT/bench.err || exit 1
$B --matmul bf16 --no-alt --cpu-steps 1 > $OUT/bench_bf16.json 2>> $OUT/bench.err || exit 1
$B --size 64 --batch 32 --sampler-steps 100 --steps 100 --warmup 10 --matmul bf16 --no-alt > $OUT/bench_c2_bf16.json 2>> $OUT/bench.err || exit 1
$B --size 64 --batch 32 --sampler-steps 100 --steps 100 --warmup 10 --matmul f32 --no-alt > $OUT/bench_c2_f32.json 2>> $OUT/bench.err || exit 1
$B --size 512 --batch 8 --steps 20 --warmup 3 --no-alt --cpu-steps 1 > $OUT/bench_c5.json 2>> $OUT/bench.err || exit 1
$B --size 512 --batch 8 --steps 20 --warmup 3 --attn fp8 --no-alt --cpu-steps 1 > $OUT/bench_c5_fp8attn.json 2>> $OUT/bench.err || exit 1
$B --size 512 --batch 8 --steps 20 --warmup 3 --matmul bf16x3 --attn fp8 --no-alt --cpu-steps 1 > $OUT/bench_c5_bf16x3_fp8attn.json 2>> $OUT/bench.err || exit 1
$B --size 512 --batch 8 --steps 20 --warmup 3 --matmul bf16x3 --no-alt --no-cpu-baseline > $OUT/bench_c5_bf16x3.json 2>> $OUT/bench.err || exit 1
$B --size 128 --batch 32 --steps 100 --warmup 10 --no-alt --no-cpu-baseline > $OUT/bench_128.json 2>> $OUT/bench.err || exit 1
$B --size 32 --batch 4 --steps 200 --warmup 20 --no-alt --cpu-steps 3 --matmul f32 > $OUT/bench_c1_gpu.json 2>> $OUT/bench.err || exit 1

step "rocprofv3 kernel traces"
prof() { name=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o run -- python3 "$@" > $OUT/prof_$name.log 2>&1) || exit 1; }
prof f16x2 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt                                   # the default command (two streams)
prof f16x2_single $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --split-streams 0          # the roofline pass's layout
prof bf16x3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --matmul bf16x3
prof f32 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --matmul f32
prof vae $ROOT/tools/vae_bench.py --iters 5
prof vae3 $ROOT/tools/vae_bench.py --iters 5 --matmul bf16x3
prof vae2 $ROOT/tools/vae_bench.py --iters 5 --matmul f16x2

step "PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes; single-stream launches as in the roofline pass)"
python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic_f16x2.json > $OUT/traffic_f16x2.txt 2>&1 || exit 1
python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic_bf16x3.json -- --matmul bf16x3 > $OUT/traffic_bf16x3.txt 2>&1 || exit 1
python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic.json -- --matmul f32 > $OUT/traffic_f32.txt 2>&1 || exit 1

step "PMC utilisation"
python3 tools/pmc_util.py --out gpurun_out/$TAG/util_f16x2.json > $OUT/util_f16x2.txt 2>&1 || exit 1
python3 tools/pmc_util.py --out gpurun_out/$TAG/util_bf16x3.json -- --matmul bf16x3 > $OUT/util_bf16x3.txt 2>&1 || exit 1
python3 tools/pmc_util.py --out gpurun_out/$TAG/util_f32.json -- --matmul f32 > $OUT/util_f32.txt 2>&1 || exit 1

step "N=2 rehearsal (gloo, both ranks on this one GPU)"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 \
    --backend gloo --share-device --no-alt > $OUT/bench_n2.out 2> $OUT/bench_n2.err || exit 1
grep '^{' $OUT/bench_n2.out > $OUT/bench_n2_gloo_rehearsal.json || exit 1

step "K sweep + in-kernel stamps of the split GEMMs"
python3 tools/micro/s3_ksweep.py > $OUT/s3_ksweep.txt 2>&1 || echo "ksweep failed"
python3 tools/micro/s3_stamps.py --build > $OUT/s3_stamps.txt 2>&1 || echo "stamps failed"

step "end to end + soak"
python3 tools/e2e_bench.py > $OUT/e2e.txt 2>&1 || echo "e2e failed"
python3 tools/soak.py > $OUT/soak.txt 2>&1 || echo "soak failed"
step done

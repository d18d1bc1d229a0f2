#!/bin/bash
# Everything the round's measurement record is built from, in one GPU-box call (run from the repo root):
#   bash tools/profile_round.sh r05
# Output under gpurun_out/<tag>/; tools/collect_profiles.py copies the judged summaries into profiles/.
# Every long step appends to a file under gpurun_out/ (gpurun kills a call that writes nothing for 7 minutes).
set -o pipefail
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*" | tee -a $OUT/progress.log; }
B="timeout -k 10 400 python3 bench.py"
C2="--size 64 --batch 32 --sampler-steps 100 --steps 100 --warmup 10 --no-alt"
C5="--size 512 --batch 8 --steps 20 --warmup 3 --no-alt"

step "bench lines"
$B > $OUT/bench.json 2> $OUT/bench.err || exit 1                                                     # default: bf16x3 headline, f32 / strict / f16x2 beside it, C3
$B --matmul f32 --no-alt > $OUT/bench_f32.json 2>> $OUT/bench.err || exit 1
$B --matmul f16x2 --no-alt > $OUT/bench_f16x2.json 2>> $OUT/bench.err || exit 1
$B --matmul bf16 --no-alt --cpu-steps 1 > $OUT/bench_bf16.json 2>> $OUT/bench.err || exit 1
step "bench lines: other configs"
$B $C2 --matmul bf16 > $OUT/bench_c2_bf16.json 2>> $OUT/bench.err || exit 1
$B $C2 --matmul f32 > $OUT/bench_c2_f32.json 2>> $OUT/bench.err || exit 1
$B $C5 --cpu-steps 1 > $OUT/bench_c5_bf16x3.json 2>> $OUT/bench.err || exit 1
$B $C5 --matmul f16x2 --no-cpu-baseline > $OUT/bench_c5_f16x2.json 2>> $OUT/bench.err || exit 1
$B $C5 --matmul f16x2 --attn fp8 --cpu-steps 1 > $OUT/bench_c5_f16x2_fp8attn.json 2>> $OUT/bench.err || exit 1
$B $C5 --attn fp8 --no-cpu-baseline > $OUT/bench_c5_bf16x3_fp8attn.json 2>> $OUT/bench.err || exit 1
$B $C2 --no-cpu-baseline > $OUT/bench_c2.json 2>> $OUT/bench.err || exit 1                             # C2 in the default (six-term bf16x3) mode
$B --size 128 --batch 32 --steps 100 --warmup 10 --no-alt --no-cpu-baseline > $OUT/bench_128.json 2>> $OUT/bench.err || exit 1
$B --size 128 --batch 16 --steps 100 --warmup 10 --no-alt --no-cpu-baseline > $OUT/bench_128_b16.json 2>> $OUT/bench.err || exit 1
$B --size 128 --batch 8 --steps 100 --warmup 10 --no-alt --no-cpu-baseline > $OUT/bench_128_b8.json 2>> $OUT/bench.err || exit 1
$B --batch 8 --steps 100 --warmup 10 --no-alt --no-cpu-baseline > $OUT/bench_c3_b8.json 2>> $OUT/bench.err || exit 1
$B --size 32 --batch 4 --steps 200 --warmup 20 --no-alt --cpu-steps 3 --matmul f32 > $OUT/bench_c1_gpu.json 2>> $OUT/bench.err || exit 1

step "round-5 A/B lines (tune keys through the environment)"
AB="$B --no-alt --no-cpu-baseline --steps 40 --warmup 5"
AVD_ATTN_M16=0 $AB > $OUT/bench_ab_attn_32x32.json 2>> $OUT/bench.err || exit 1          # the 32x32x16 attention pipeline (round 4's shape, peeled)
AVD_ATTN_M16=0 AVD_ATTN_PIPE=0 $AB > $OUT/bench_ab_attn_plain.json 2>> $OUT/bench.err || exit 1   # the round-3 attention kernel
AVD_S3_SN=16 AVD_S3_SUPER4=32 $AB > $OUT/bench_ab_supertile_r4.json 2>> $OUT/bench.err || exit 1   # rounds 2-4's block order: 2 block rows x all columns
AVD_CFG_ROWS=0 $AB > $OUT/bench_ab_cfg_gather.json 2>> $OUT/bench.err || exit 1          # CFG + un-patch + DDIM as one 16-byte gather per lane
AVD_MLP_FUSED=1 $AB > $OUT/bench_ab_mlp_fused.json 2>> $OUT/bench.err || exit 1          # fc1 -> GELU -> fc2 as one launch (round 4; kept as a record)
$AB > $OUT/bench_ab_default.json 2>> $OUT/bench.err || exit 1                            # the default, same flags, right after
AVD_ATTN_M16=0 $B $C5 --no-cpu-baseline > $OUT/bench_ab_c5_attn_32x32.json 2>> $OUT/bench.err || exit 1

step "rocprofv3 kernel traces"
prof() { name=$1; shift; (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o run -- python3 "$@" > $OUT/prof_$name.log 2>&1) || exit 1; step "  traced $name"; }
prof bf16x3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt                                  # the default command (headline mode)
prof f16x2 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --matmul f16x2                    # speed mode, its two-stream default
prof f16x2_single $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --matmul f16x2 --split-streams 0
prof f32 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --matmul f32
prof vae $ROOT/tools/vae_bench.py --iters 5
prof vae3 $ROOT/tools/vae_bench.py --iters 5 --matmul bf16x3
prof vae2 $ROOT/tools/vae_bench.py --iters 5 --matmul f16x2
prof vae3enc $ROOT/tools/vae_bench.py --iters 5 --matmul bf16x3 --encode

step "PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes; single-stream launches as in the roofline pass)"
timeout -k 10 600 python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic_bf16x3.json -- --matmul bf16x3 > $OUT/traffic_bf16x3.txt 2>&1 || exit 1
step "  traffic bf16x3"
AVD_MLP_FUSED=1 timeout -k 10 600 python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic_bf16x3_mlpfused.json -- --matmul bf16x3 > $OUT/traffic_bf16x3_mlpfused.txt 2>&1 || echo "fused traffic failed"
step "  traffic bf16x3, fused MLP"
timeout -k 10 600 python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic_f16x2.json -- --matmul f16x2 > $OUT/traffic_f16x2.txt 2>&1 || exit 1
step "  traffic f16x2"
timeout -k 10 600 python3 tools/pmc_traffic.py --out gpurun_out/$TAG/traffic.json -- --matmul f32 > $OUT/traffic_f32.txt 2>&1 || exit 1

step "PMC utilisation"
timeout -k 10 600 python3 tools/pmc_util.py --out gpurun_out/$TAG/util_bf16x3.json -- --matmul bf16x3 > $OUT/util_bf16x3.txt 2>&1 || exit 1
step "  util bf16x3"
timeout -k 10 600 python3 tools/pmc_util.py --out gpurun_out/$TAG/util_f16x2.json -- --matmul f16x2 > $OUT/util_f16x2.txt 2>&1 || exit 1
step "  util f16x2"
timeout -k 10 600 python3 tools/pmc_util.py --out gpurun_out/$TAG/util_f32.json -- --matmul f32 > $OUT/util_f32.txt 2>&1 || exit 1

step "N=2 rehearsal (gloo, both ranks on this one GPU)"
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 \
    --backend gloo --share-device --no-alt --verify-ranks > $OUT/bench_n2.out 2> $OUT/bench_n2.err || exit 1
grep '^{' $OUT/bench_n2.out > $OUT/bench_n2_gloo_rehearsal.json || exit 1

step "per-kernel clock / power, block phases of fc1 / in_proj, L2 counters"
timeout -k 10 300 python3 tools/micro/kernel_power.py > $OUT/kernel_power.txt 2>&1 || echo "kernel_power failed"
timeout -k 10 400 python3 tools/pmc_counters.py --out gpurun_out/$TAG/l2_counters_bf16x3.json --sets "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" > $OUT/l2_counters_bf16x3.txt 2>&1 || echo "l2 counters failed"
timeout -k 10 500 python3 tools/micro/s3_phase.py > $OUT/s3_phase.txt 2>&1 || echo "s3_phase failed"
step "K sweep + in-kernel stamps of the split GEMMs"
timeout -k 10 300 python3 tools/micro/s3_ksweep.py > $OUT/s3_ksweep.txt 2>&1 || echo "ksweep failed"
timeout -k 10 400 python3 tools/micro/s3_stamps.py > $OUT/s3_stamps.txt 2>&1 || echo "stamps failed"
for v in AVD_LAB_NODMA AVD_LAB_NOLDS AVD_LAB_NOSTORE; do
    echo "== variant $v (diagnostic build, wrong results by design)" >> $OUT/s3_stamps.txt
    timeout -k 10 400 python3 tools/micro/s3_stamps.py --modes bf16x3,f16x2 --variant $v >> $OUT/s3_stamps.txt 2>&1 || echo "stamps $v failed"
done

step "VAE decode timings, end to end, soak"
for m in f32 bf16x3 f16x2; do timeout -k 10 200 python3 tools/vae_bench.py --matmul $m --power 3 2>&1 | grep -v amdgpu.ids >> $OUT/vae_decode.txt; done
for m in bf16x3 f16x2; do AVD_VAE_FOLD=0 timeout -k 10 200 python3 tools/vae_bench.py --matmul $m --power 3 2>&1 | grep -v amdgpu.ids | sed "s/^\[$m\]/[$m, fold 0]/" >> $OUT/vae_decode.txt; done      # fp32 activations between the kernels
for m in bf16x3 f16x2; do AVD_VAE_FOLD=0 timeout -k 10 200 python3 tools/vae_bench.py --matmul $m --lat 0 2>&1 | grep -v amdgpu.ids | sed "s/, lat 0\]/, lat 0, fold 0]/" >> $OUT/vae_decode.txt; done      # rounds 1-4: 64-channel first conv
timeout -k 10 200 python3 tools/vae_bench.py --matmul bf16x3 --batch 8 2>&1 | grep -v amdgpu.ids >> $OUT/vae_decode.txt
timeout -k 10 300 python3 tools/pmc_util.py --script tools/vae_bench.py --out gpurun_out/$TAG/util_vae3.json -- --matmul bf16x3 --iters 2 > $OUT/util_vae3.txt 2>&1 || echo "vae util failed"
timeout -k 10 200 python3 tools/vae_bench.py --matmul f16x2 --batch 8 2>&1 | grep -v amdgpu.ids >> $OUT/vae_decode.txt
for m in f32 bf16x3 f16x2; do timeout -k 10 200 python3 tools/vae_bench.py --encode --matmul $m 2>&1 | grep -v amdgpu.ids >> $OUT/vae_encode.txt; done
for m in bf16x3 f16x2; do AVD_VAE_FOLD=0 timeout -k 10 200 python3 tools/vae_bench.py --encode --matmul $m 2>&1 | grep -v amdgpu.ids | sed "s/^\[$m\]/[$m, fold 0]/" >> $OUT/vae_encode.txt; done
timeout -k 10 600 python3 tools/e2e_bench.py > $OUT/e2e.txt 2>&1 || echo "e2e failed"
step "  e2e done"
timeout -k 10 600 python3 tools/soak.py > $OUT/soak.txt 2>&1 || echo "soak failed"
step done

#!/bin/bash
# Round 5, GPU call A: the evidence VERDICT r4 item 1 asks for before any fc1 / in_proj change.
#   1. tools/micro/s3_phase.py: per-block XCD / time stamps -> phase picture + trace-driven L2 model; AVD_LAB_ALIAS builds = the time the
#      kernels would take with every staged piece an L2 hit (upper bound of what panel residency can buy)
#   2. TCC hit / miss / request counters per kernel of the default bench step
#   3. FETCH_SIZE / WRITE_SIZE of the step with the first-generation stagger off and across super-tile shapes
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5a; mkdir -p $OUT; export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*" | tee -a $OUT/progress.log; }
step "s3_phase"
timeout -k 10 500 python3 tools/micro/s3_phase.py --alias > $OUT/s3_phase.txt 2>&1 || { step "s3_phase failed"; tail -5 $OUT/s3_phase.txt; }
step "TCC counters"
timeout -k 10 400 python3 tools/pmc_counters.py --out gpurun_out/r5a/l2_bf16x3.json --sets "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" > $OUT/l2_bf16x3.txt 2>&1 || step "counters failed"
step "fetch A/B"
ab() { name=$1; shift; env "$@" timeout -k 10 400 python3 tools/pmc_traffic.py --out gpurun_out/r5a/traffic_$name.json > $OUT/traffic_$name.txt 2>&1 || step "traffic $name failed"; step "  traffic $name"; }
ab default AVD_NOP=1
ab stagger0 AVD_S3_STAGGER=0
ab super16 AVD_S3_SUPER4=16
ab super64 AVD_S3_SUPER4=64
ab sn8_super32 AVD_S3_SN=8
ab sn4_super32 AVD_S3_SN=4
step done

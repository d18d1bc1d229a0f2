#!/bin/bash
# Round 5, GPU call B: the GPU test tier on the round's first changes (sharded decode, header broadcast) + the times of the block-order variants
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5b; mkdir -p $OUT; export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*" | tee -a $OUT/progress.log; }
step "supertile A/B"
timeout -k 10 400 python3 tools/micro/s3_supertile_ab.py > $OUT/supertile_ab.txt 2>&1 || step "supertile ab failed"
step "pytest gpu"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.log
tail -5 $OUT/pytest_gpu.txt
step done

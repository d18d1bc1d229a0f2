#!/bin/bash
# Round 5, GPU call C: block-order A/B again on the exact (unpadded) grids, incl. the real in_proj epilogue; the new parity tests
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5c; mkdir -p $OUT; export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*" | tee -a $OUT/progress.log; }
step "supertile A/B"
timeout -k 10 400 python3 tools/micro/s3_supertile_ab.py > $OUT/supertile_ab.txt 2>&1 || step "supertile ab failed"
timeout -k 10 400 python3 tools/micro/s3_supertile_ab.py --rows 8512 > $OUT/supertile_ab_8512.txt 2>&1 || step "supertile ab 8512 failed"
step "pytest new tests"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "shipped_config or chain_full_width or end_to_end or two_stream or stream_generate or split_gemm or tile_configurations" > $OUT/pytest_new.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.log
tail -15 $OUT/pytest_new.txt
step "bench"
timeout -k 10 300 python3 bench.py --no-alt > $OUT/bench.json 2> $OUT/bench.err || step "bench failed"
step done

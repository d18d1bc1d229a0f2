#!/bin/bash
# Round 5, GPU call F: latent-composed first decoder conv: VAE tests, decode timings both ways; CFG row kernel after the contraction fix
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r5f; mkdir -p $OUT; export TMPDIR=/tmp
step() { echo "[$(date +%T)] $*" | tee -a $OUT/progress.log; }
step "pytest vae + cfg"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "vae or cfg_unpatch or end_to_end or shipped_config or stream_generate or golden" > $OUT/pytest_sel.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.log
tail -25 $OUT/pytest_sel.txt
step "vae bench"
for m in bf16x3 f16x2; do for lat in 0 1; do timeout -k 10 200 python3 tools/vae_bench.py --matmul $m --lat $lat 2>&1 | grep -v amdgpu.ids >> $OUT/vae_decode.txt; done; done
timeout -k 10 200 python3 tools/vae_bench.py --matmul f16x2 --batch 8 2>&1 | grep -v amdgpu.ids >> $OUT/vae_decode.txt
timeout -k 10 200 python3 tools/vae_bench.py --matmul bf16x3 --batch 8 2>&1 | grep -v amdgpu.ids >> $OUT/vae_decode.txt
cat $OUT/vae_decode.txt
step done

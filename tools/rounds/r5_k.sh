#!/bin/bash
# round-5 GPU call K: the folded VideoVAE decoder route (conv 0 -> operand image, GroupNorm folded into conv 1, to_img from partial sums) and
# the two-launch gn_finalize: VAE parity tests, per-kernel decode timings with the fold on / off, clock / power of the decode loop.
set -o pipefail
OUT=gpurun_out/r5k
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] VAE tests"
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_f16x2.py -m gpu -x -q -k "vae or sample_one_direction or stream" > $OUT/pytest.txt 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest.txt
echo "[$(date +%T)] decode timings"
for m in bf16x3 f16x2; do timeout -k 10 200 python3 tools/vae_bench.py --matmul $m --power 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vae_decode.txt; done
AVD_VAE_FOLD=0 timeout -k 10 200 python3 tools/vae_bench.py --matmul bf16x3 --power 3 2>&1 | grep -v amdgpu.ids | sed 's/^\[bf16x3\]/[bf16x3, fold 0]/' | tee -a $OUT/vae_decode.txt
timeout -k 10 200 python3 tools/vae_bench.py --matmul bf16x3 --lat 0 --power 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vae_decode.txt
timeout -k 10 200 python3 tools/vae_bench.py --matmul bf16x3 --batch 8 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vae_decode.txt
timeout -k 10 200 python3 tools/vae_bench.py --matmul f32 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vae_decode.txt
echo "[$(date +%T)] done"

#!/bin/bash
# round-4 GPU call Q: the three-plane conv on 8-row tiles (128-voxel wave tile, 103 KB halo tile, one block per CU, accumulators in AGPRs by the
# compiler) against the default 4-row tiles, two blocks per CU.  Result (profiles/r04_vae_conv_th8.txt): correct, not faster.
set -o pipefail
OUT=gpurun_out/r4q
mkdir -p $OUT
export TMPDIR=/tmp
C=multimodal_diffusion_amd/csrc
echo "[$(date +%T)] default build"
timeout -k 10 200 python3 tools/vae_bench.py --matmul bf16x3 2>&1 | grep -v amdgpu.ids | tee $OUT/vae_default.txt
# the diagnostic build edits a tracked source and relinks the product library: both are restored when the script ends, however it ends
cp $C/vae3d_f32.hip $OUT/vae3d_f32.hip.orig && cp $C/libavdiff_hip.so $OUT/libavdiff_hip.so.orig && cp $C/vae3d_f32.o $OUT/vae3d_f32.o.orig || exit 1
trap 'cp $OUT/vae3d_f32.hip.orig $C/vae3d_f32.hip; cp $OUT/vae3d_f32.o.orig $C/vae3d_f32.o; cp $OUT/libavdiff_hip.so.orig $C/libavdiff_hip.so; rm -f $OUT/*.orig' EXIT
echo "[$(date +%T)] rebuild vae3d_f32.o with TH = 8 for three planes, one block per CU (edits the box's scratch copy only)"
sed -i 's/static constexpr int TH = NPL == 2 ? 8 : 4, HH = TH + 2;/static constexpr int TH = 8, HH = TH + 2;/; s/__global__ __launch_bounds__(256, 2) void conv3d_k3_bf16x3_kernel/__global__ __launch_bounds__(256, TERMS == 3 ? 2 : 1) void conv3d_k3_bf16x3_kernel/' $C/vae3d_f32.hip
grep -c "static constexpr int TH = 8, HH" $C/vae3d_f32.hip || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -c $C/vae3d_f32.hip -o $C/vae3d_f32.o || exit 1
(cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libavdiff_hip.so gemm_f32.o gemm_bf16x3.o mlp_bf16x3.o attn_f32.o attn_bf16x3.o attn_fp8.o rowops.o tokens.o vae3d_f32.o codec_f32.o stitch.o composite.o) || exit 1
timeout -k 10 200 python3 tools/vae_bench.py --matmul bf16x3 2>&1 | grep -v amdgpu.ids | tee $OUT/vae_th8.txt
timeout -k 10 200 python3 tools/vae_bench.py --matmul bf16x3 --batch 8 2>&1 | grep -v amdgpu.ids | tee -a $OUT/vae_th8.txt
echo "[$(date +%T)] VAE tests on the diagnostic build"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_f16x2.py -m gpu -q -k "vae" 2>&1 | tail -3
echo "[$(date +%T)] done"

#!/bin/bash
# round-4 GPU call E: fused MLP parity, trim / split-K subset; benches
set -o pipefail
OUT=gpurun_out/r4e
mkdir -p $OUT
export TMPDIR=/tmp
echo "[$(date +%T)] pytest fused MLP"
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -s -k "fused_mlp" > $OUT/pytest_mlp.log 2>&1 || { tail -40 $OUT/pytest_mlp.log; echo FUSED-MLP-TEST-FAILED; }
grep -E "fused MLP|passed|failed" $OUT/pytest_mlp.log | tail -5
echo "[$(date +%T)] pytest subset"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "last_block or block_rows or splitk or shipped or full_step_c3 or tile_configurations or chain or default_mode" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; echo "pytest failed: no bench lines from a failing tree"; exit 1; }
tail -3 $OUT/pytest.log
B="timeout -k 10 300 python3 bench.py --no-alt --no-cpu-baseline"
for p in 0 1; do
  AVD_MLP_FUSED=$p $B --steps 30 --warmup 5 > $OUT/bench_mlp$p.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
  echo "C3 mlp_fused=$p:"; python3 tools/bench_kernels.py $OUT/bench_mlp$p.json > $OUT/k.txt; head -8 $OUT/k.txt
done
for p in 0 4; do
  AVD_S3_SPLITK=$p $B --size 128 --batch 32 --steps 100 --warmup 10 > $OUT/bench_128_sk$p.json 2>> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
  echo "128 splitk=$p:"; python3 tools/bench_kernels.py $OUT/bench_128_sk$p.json > $OUT/k.txt; head -9 $OUT/k.txt
done
AVD_S3_SPLITK=4 AVD_CORE_TRIM=0 $B --size 128 --batch 32 --steps 100 --warmup 10 > $OUT/bench_128_sk4_t0.json 2>> $OUT/bench.err || exit 1
echo "128 splitk=4 trim=0: $(python3 tools/bench_kernels.py $OUT/bench_128_sk4_t0.json | head -1)"
for p in 0 1; do
  AVD_CORE_TRIM=$p $B --steps 30 --warmup 5 > $OUT/bench_t$p.json 2>> $OUT/bench.err || exit 1
  echo "C3 core_trim=$p: $(python3 tools/bench_kernels.py $OUT/bench_t$p.json | head -1)"
done
$B --size 512 --batch 8 --steps 20 --warmup 3 > $OUT/bench_c5.json 2>> $OUT/bench.err || exit 1
echo "C5: $(python3 tools/bench_kernels.py $OUT/bench_c5.json | head -1)"
echo "[$(date +%T)] done"

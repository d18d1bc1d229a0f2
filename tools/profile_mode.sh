#!/bin/bash
# Kernel trace + utilisation counters of one matmul mode's bench step (GPU box, repo root):  bash tools/profile_mode.sh f16x2 tag
MODE=${1:-f16x2}; TAG=${2:-r02q}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python3 bench.py --matmul $MODE --no-alt --cpu-steps 1 --steps 100 > $OUT/bench_$MODE.json 2> $OUT/bench_$MODE.err || exit 1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$MODE -o run -- python3 $ROOT/bench.py --matmul $MODE --steps 10 --warmup 3 --no-cpu-baseline --no-alt > $OUT/prof_$MODE.log 2>&1) || exit 1
python3 tools/pmc_util.py --out gpurun_out/$TAG/util_$MODE.json -- --matmul $MODE > $OUT/util_$MODE.txt 2>&1 || exit 1

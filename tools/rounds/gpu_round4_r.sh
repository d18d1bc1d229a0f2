#!/bin/bash
# round-4 GPU call R: rocprofv3 kernel traces of the mid-size configurations (default mode) and of 128x128 with round 3's blocks
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r4r
mkdir -p $OUT
export TMPDIR=/tmp
prof() { name=$1; shift; (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o run -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-roofline "$@" > $OUT/prof_$name.log 2>&1) || exit 1; echo "traced $name"; }
prof 128 --size 128
prof c2 --size 64 --batch 32
prof 128_b8 --size 128 --batch 8
export AVD_S3_RT4=8 AVD_S3_DEEP4=0
prof 128_round3_blocks --size 128
ls $OUT/prof_128

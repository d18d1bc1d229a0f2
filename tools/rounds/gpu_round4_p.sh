#!/bin/bash
# round-4 GPU call P: in-kernel stamps of the short four-wave blocks (rows 3,904 -> RT 2, rows 8,512 -> RT 5, four-stage ring), with the
# diagnostic builds that drop the in-loop DMA / the fragment reads
set -o pipefail
OUT=gpurun_out/r4p
mkdir -p $OUT
export TMPDIR=/tmp
for v in "" AVD_LAB_NODMA AVD_LAB_NOLDS; do
  for rows in 3904 8512; do
    echo "== variant '$v' rows $rows" >> $OUT/stamps.txt
    timeout -k 10 400 python3 tools/micro/s3_stamps.py --modes bf16x3 --rows $rows ${v:+--variant $v} 2>&1 | grep -v amdgpu.ids >> $OUT/stamps.txt || echo "failed $v $rows" >> $OUT/stamps.txt
  done
done
cut -c1-230 $OUT/stamps.txt

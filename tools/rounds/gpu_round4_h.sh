#!/bin/bash
# round-4 GPU call H: the whole GPU test tier + smoke, as the driver runs them
set -o pipefail
OUT=gpurun_out/r4h
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1 || { grep -E "FAILED|Error" $OUT/pytest.log | head -30; tail -5 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

#!/bin/bash
# usage: tools/gpu_sweep.sh ENVVAR "v1 v2 ..." "mode1 mode2" [extra bench args]   — one steps/s line per (value, mode)
V=$1; VALS=$2; MODES=${3:-"bf16x3 f16x2"}; shift $(( $# < 3 ? $# : 3 ))
for m in $MODES; do for v in $VALS; do
  r=$(env $V=$v timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-alt --no-cpu-baseline --no-roofline --matmul $m "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "$m $V=$v -> $r steps/s"
done; done

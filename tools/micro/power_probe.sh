#!/bin/bash
# Clock and socket power while bench.py loops in one matrix-pipe mode (GPU box): bash tools/micro/power_probe.sh [modes...]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/power
for m in ${@:-f16x2 bf16x3 f32}; do
  python3 bench.py --matmul $m --no-alt --no-cpu-baseline --no-roofline --steps 6000 > gpurun_out/power/bench_$m.json 2>/dev/null &
  BP=$!
  : > gpurun_out/power/smi_$m.txt
  n=0
  for i in $(seq 1 150); do
    line=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' ')
    mhz=$(echo "$line" | sed -n 's/.*sclk clock level: [^(]*(\([0-9]*\)Mhz).*/\1/p')
    if [ -n "$mhz" ] && [ "$mhz" -gt 1000 ]; then echo "$line" >> gpurun_out/power/smi_$m.txt; n=$((n+1)); fi
    [ $n -ge 8 ] && break
    kill -0 $BP 2>/dev/null || break
    sleep 1
  done
  wait $BP
done
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|cap" | head -4 > gpurun_out/power/cap.txt

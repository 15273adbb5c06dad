#!/bin/bash
# Sample rocm-smi clocks/power while the benchmark loop runs (evidence for the DVFS discussion in DESIGN.md).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python bench.py --steps 400 --warmup 5 --no-cpu-baseline > gpurun_out/clock_bench.json 2>&1) &
BP=$!
n=0
while kill -0 $BP 2>/dev/null && [ $n -lt 12 ]; do
  s=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | tr -s ' \t' ' ' | sed 's/GPU\[0\] : //' | tr '\n' '|')
  case "$s" in *"(95Mhz)"*|*"(132Mhz)"*) ;; *) echo "$s"; n=$((n+1));; esac
  sleep 0.7
done
wait $BP
tail -c 300 gpurun_out/clock_bench.json

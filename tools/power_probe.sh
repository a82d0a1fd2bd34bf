#!/bin/bash
# GPU box: socket power and shader clock sampled by rocm-smi while one conv shape runs in a loop.
# usage: power_probe.sh <tag> <DATA=rand|zeros> <bench_conv.py args...>
tag=$1; data=$2; shift 2
DATA=$data python tools/bench_conv.py "$@" > gpurun_out/power_${tag}.log 2>&1 &
pid=$!
sleep 12   # import + warm-up
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk|Average Graphics Package Power|Current Socket" | tr '\n' ' '
  echo
  sleep 1
done
wait $pid
grep -v amdgpu gpurun_out/power_${tag}.log

#!/bin/bash
# feasibility probe for the tied nearest-x2 decomposition: a 3x3 conv over the upsampled map against the 4-phase ConvTranspose2d(k4,s2,p1)
# over the half-resolution map (same output grid, 4/9 of the MACs), per class, through the single-op entry points
set -e
out=gpurun_out/tied_probe.txt; : > $out
for shape in "88 512 256" "176 256 128" "44 1024 256" "176 512 128"; do
  set -- $shape; h=$1; ci=$2; co=$3
  for mode in fwd dgrad wgrad; do
    python tools/bench_conv.py 16 $((h*2)) $((h*2)) $ci $co 3 1 $mode 10 >> $out
    TR=1 python tools/bench_conv.py 16 $h $h $ci $co 4 2 $mode 10 >> $out
  done
done
cat $out

#!/bin/bash
# 4-tap ConvTranspose2d launches: 16x16 against 8x16 pixel tiles (two workgroups per CU)
out=gpurun_out/tied_probe2.txt; : > $out
for shape in "88 512 256" "176 256 128" "176 512 128" "176 128 64"; do
  set -- $shape; h=$1; ci=$2; co=$3
  for wm in 0 2 4; do
    echo "FORCE_WM=$wm" >> $out
    OCTSEG_FORCE_WM=$wm TR=1 python tools/bench_conv.py 16 $h $h $ci $co 4 2 fwd 10 >> $out
  done
done
cat $out

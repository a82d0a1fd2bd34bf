#!/bin/bash
# 16-tap stride-2 data gradient of ConvTranspose2d (the tied data gradient's launch): K-chunk / tile variants
out=gpurun_out/tied_probe4.txt; : > $out
for shape in "88 512 256" "176 256 128" "176 512 128" "176 128 64"; do
  set -- $shape; h=$1; ci=$2; co=$3
  for v in "base" "rb64" "rb64wm4"; do
    unset OCTSEG_S2_RB64 OCTSEG_FORCE_WM
    [ $v != base ] && export OCTSEG_S2_RB64=1
    [ $v = rb64wm4 ] && export OCTSEG_FORCE_WM=4
    echo -n "$v: " >> $out
    TR=1 python tools/bench_conv.py 16 $h $h $ci $co 4 2 dgrad 10 >> $out
  done
done
cat $out

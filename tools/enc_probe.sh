#!/bin/bash
# ResNet encoder 3x3 shapes at 16 frames: 11 x 11 pixel tiles (LOOP_T11) against the 16-pixel tilings (OCTSEG_NO_TILE11) and their variants
out=gpurun_out/enc_probe.txt; : > $out
for shape in "44 256 256" "88 128 128" "22 512 512" "44 3072 256" "88 512 512"; do
  set -- $shape; h=$1; ci=$2; co=$3
  for mode in fwd dgrad; do
    for v in t11 base; do
      unset OCTSEG_NO_TILE11
      [ $v != t11 ] && export OCTSEG_NO_TILE11=1
      echo -n "$v: " >> $out
      python tools/bench_conv.py 16 $h $h $ci $co 3 1 $mode 20 >> $out
    done
  done
done
cat $out

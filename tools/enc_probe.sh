#!/bin/bash
# ResNet encoder 3x3 shapes at 16 frames: tile variants against the round quantisation (288 tiles on 256 CUs at 44^2)
out=gpurun_out/enc_probe.txt; : > $out
for shape in "44 256 256" "88 128 128" "22 512 512"; do
  set -- $shape; h=$1; ci=$2; co=$3
  for mode in fwd dgrad; do
    for v in base wm2 wm4 n256; do
      unset OCTSEG_FORCE_WM OCTSEG_N256_ALL OCTSEG_NO_SMALLGRID
      [ $v = wm2 ] && export OCTSEG_FORCE_WM=2
      [ $v = wm4 ] && export OCTSEG_FORCE_WM=4
      [ $v = n256 ] && export OCTSEG_N256_ALL=1
      echo -n "$v: " >> $out
      python tools/bench_conv.py 16 $h $h $ci $co 3 1 $mode 20 >> $out
    done
  done
done
cat $out

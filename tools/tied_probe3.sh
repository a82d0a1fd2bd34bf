#!/bin/bash
# ConvTranspose2d weight gradients: one fused 16-tap launch (wgrad_convt.hip) against four per-parity launches (window-row-major 4-tap loop,
# and the single-buffer loop before it)
out=gpurun_out/tied_probe3.txt; : > $out
for shape in "88 512 256" "176 256 128" "176 512 128" "176 128 64" "44 1024 256" "176 64 32"; do
  set -- $shape; h=$1; ci=$2; co=$3
  for sw in fused parity4 parity4-single; do
    unset OCTSEG_NO_WGRAD_PIPE2_4 OCTSEG_NO_WGRAD_CONVT16
    [ $sw != fused ] && export OCTSEG_NO_WGRAD_CONVT16=1
    [ $sw = parity4-single ] && export OCTSEG_NO_WGRAD_PIPE2_4=1
    echo -n "$sw: " >> $out
    TR=1 python tools/bench_conv.py 16 $h $h $ci $co 4 2 wgrad 10 >> $out
  done
done
cat $out

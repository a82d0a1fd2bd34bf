#!/bin/bash
# GPU box: the round-4 workloads the first collection did not reach + the co-tile-order A/B with TCC traffic
cd /root/repo; export TMPDIR=/tmp
for w in pan_r50_704 unet_regnetx064_704 unet_regnety120_704 fpn_regnetx002_704 unet_effb0_704 fpn_effb5_704; do
  python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench_$w.json 2> gpurun_out/r4_bench_$w.err || tail -3 gpurun_out/r4_bench_$w.err
done
bash tools/r3_ab.sh r4co "-" "OCTSEG_WGRAD_CI_MAJOR=1" "-" "OCTSEG_WGRAD_CI_MAJOR=1" > gpurun_out/r4_wgrad_co_order_ab.txt 2>&1
python tools/collect_traffic.py r4 > gpurun_out/r4_traffic.log 2>&1
tail -2 gpurun_out/r4_traffic.log

#!/bin/bash
# GPU box (under gpurun): the round's committed measurements.  Writes under gpurun_out/ (copy into profiles/ afterwards).
# usage: bash tools/collect_profiles.sh <tag>
set -e
tag=${1:-r1}
part=${2:-all}      # core | workloads | all (two gpurun calls when the whole list does not fit one call's time limit)
cd /root/repo; export TMPDIR=/tmp
if [ "$part" != "workloads" ]; then
python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_w -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${tag}_w.log 2>&1
python3 tools/prof_summary.py $(ls gpurun_out/prof_${tag}_w/*/*.db | head -1) gpurun_out/${tag}_w_bench_kernel_stats.csv 8 > gpurun_out/prof_${tag}_w.txt
OCTSEG_NO_SIDE_STREAM=1 OCTSEG_NO_FWD_LANES=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_serial -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${tag}_serial.log 2>&1
python3 tools/prof_summary.py $(ls gpurun_out/prof_${tag}_serial/*/*.db | head -1) gpurun_out/${tag}_serial_kernel_stats.csv 8 > gpurun_out/prof_${tag}_serial.txt
python3 tools/collect_traffic.py ${tag} > gpurun_out/${tag}_traffic.log 2>&1
rm -rf gpurun_out/prof_${tag}_w gpurun_out/prof_${tag}_serial gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
OCTSEG_PROFILE_DUMP=gpurun_out/${tag}_layers_alone.csv python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 tools/group_layers.py gpurun_out/${tag}_layers_alone.csv 2 > gpurun_out/${tag}_layer_groups.txt
python3 bench.py --workload ensemble_704_fp16 --steps 30 > gpurun_out/${tag}_ensemble_b1.json 2> /dev/null
python3 bench.py --workload ensemble_704_fp16 --steps 20 --batch 8 > gpurun_out/${tag}_ensemble_b8.json 2> /dev/null
fi
if [ "$part" != "core" ]; then
for w in linknet_r50_704 unet_r50_704 fpn_r50_704 deeplabv3plus_r50_704 pspnet_r50_704 deeplabv3_r50_704 manet_r50_704 pan_r50_704 unet_regnetx064_704 unet_regnety120_704 \
         fpn_regnetx002_704 unet_effb0_704 fpn_effb5_704; do
  OCTSEG_PROFILE_DUMP=gpurun_out/${tag}_layers_$w.csv python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_$w.json 2> /dev/null
  python3 tools/group_layers.py gpurun_out/${tag}_layers_$w.csv 2 > gpurun_out/${tag}_layer_groups_$w.txt; rm -f gpurun_out/${tag}_layers_$w.csv
done
for b in 2 4; do python3 bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_batch$b.json 2> /dev/null; done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-exchange > gpurun_out/${tag}_bench_force_exchange.json 2> /dev/null
python3 bench.py --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --train-graph > gpurun_out/${tag}_bench_batch2_train_graph.json 2> /dev/null
tail -c 600 gpurun_out/${tag}_bench_default.err
fi

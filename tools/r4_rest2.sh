#!/bin/bash
# GPU box: group-wise XCD placement of the weight-gradient workgroups: ops parity, ABAB with the kernels-alone table, TCC traffic; then the
# small-batch / graph / exchange lines of the round
cd /root/repo; export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
bash tools/r3_ab.sh r4xg "-" "OCTSEG_WGRAD_CI_MAJOR=1" "-" "OCTSEG_WGRAD_CI_MAJOR=1" > gpurun_out/r4_wgrad_xcd_groups_ab.txt 2>&1
python tools/collect_traffic.py r4 > gpurun_out/r4_traffic.log 2>&1
tail -1 gpurun_out/r4_traffic.log
for b in 2 4; do python3 bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_bench_batch$b.json 2> /dev/null; done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-exchange > gpurun_out/r4_bench_force_exchange.json 2> /dev/null
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-exchange --allreduce-dtype bf16 > gpurun_out/r4_bench_force_exchange_bf16.json 2> /dev/null
python3 bench.py --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --train-graph > gpurun_out/r4_bench_batch2_train_graph.json 2> /dev/null
OCTSEG_NO_SIDE_STREAM=1 OCTSEG_NO_FWD_LANES=1 python3 bench.py --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --train-graph > gpurun_out/r4_bench_batch2_train_graph_1stream.json 2> /dev/null
OCTSEG_NO_SIDE_STREAM=1 OCTSEG_NO_FWD_LANES=1 python3 bench.py --batch 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_bench_batch2_1stream.json 2> /dev/null
for f in batch2 batch4 force_exchange force_exchange_bf16 batch2_train_graph batch2_train_graph_1stream batch2_1stream; do python3 -c "
import json
d=json.load(open('gpurun_out/r4_bench_$f.json')); print('%-32s %8.1f frames/s %7.2f ms/step' % ('$f', d['value'], d['ms_per_step']))"; done

# GPU box: GPU_MAX_HW_QUEUES (HIP's hardware-queue pool) against graph replay and the eager multi-stream step, one run
for q in 4 8 2; do
  for args in "--batch 2 --steps 20 --warmup 5" "--batch 2 --steps 20 --warmup 5 --train-graph" "--steps 6 --warmup 2" "--steps 6 --warmup 2 --train-graph" "--steps 6 --warmup 2 --force-exchange" "--workload ensemble_704_fp16 --steps 20 --batch 1"; do
    GPU_MAX_HW_QUEUES=$q python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues=$q', '$args', d['value'], d['ms_per_step'])"
  done
done

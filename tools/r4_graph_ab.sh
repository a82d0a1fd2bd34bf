run() { python bench.py "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print(o['value'], o['ms_per_step'])
"; }
echo "ensemble b1 default"; run --workload ensemble_704_fp16 --steps 30
echo "ensemble b1 prio=low"; OCTSEG_SIDE_PRIORITY=low run --workload ensemble_704_fp16 --steps 30
echo "b2 graph default"; run --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --train-graph
echo "b2 graph prio=normal"; OCTSEG_SIDE_PRIORITY=normal run --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --train-graph
echo "b2 graph tied=0"; OCTSEG_TIED=0 run --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --train-graph
echo "b2 graph tied=0 prio=normal"; OCTSEG_TIED=0 OCTSEG_SIDE_PRIORITY=normal run --batch 2 --steps 20 --warmup 5 --no-cpu-baseline --train-graph
echo "b2 eager"; run --batch 2 --steps 20 --warmup 5 --no-cpu-baseline
echo "default"; run --no-cpu-baseline

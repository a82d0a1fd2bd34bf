#!/bin/bash
# GPU box: quick parity gate + per-layer profile of the benchmark step.  usage: tools/r2_perf.sh <tag>
set -e
tag=$1
python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py -m gpu -q -x > gpurun_out/${tag}_gate.log 2>&1 || { tail -30 gpurun_out/${tag}_gate.log; exit 1; }
tail -2 gpurun_out/${tag}_gate.log
OCTSEG_PROFILE_DUMP=gpurun_out/layers_${tag}.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_bench.json'))
r = d['roofline']
print('frames/s', d['value'], 'ms/step', d['ms_per_step'], 'mfma alone ms', r['kernel_ms_per_step'], 'TF/s', r['achieved'], {k: v['ms_per_step'] for k, v in r['by_class'].items()}, 'hbm sweeps ms', d.get('roofline_hbm', {}).get('kernel_ms_per_step'))
PY
python tools/group_layers.py gpurun_out/layers_${tag}.csv 2

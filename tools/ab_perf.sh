#!/bin/bash
# GPU box: A/B of environment switches inside ONE run (box-to-box spread is +-3 %).  usage: ab_perf.sh "<env A>" "<env B>" [more...]
i=0
for e in "$@"; do
  i=$((i+1))
  env $e OCTSEG_PROFILE_DUMP=gpurun_out/layers_ab$i.csv python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/ab$i.json 2>/dev/null
  echo "== [$e]"; python - <<PY
import json
d = json.load(open('gpurun_out/ab$i.json')); r = d['roofline']
print('frames/s', d['value'], 'mfma alone', r['kernel_ms_per_step'], {k: v['ms_per_step'] for k, v in r['by_class'].items()})
PY
  python tools/group_layers.py gpurun_out/layers_ab$i.csv 2 | grep -E "decoder 3x3|encoder 3x3 .*(fwd|dgrad)"
done

#!/bin/bash
# GPU box: A/B of environment switches inside ONE run, with the per-group table of the kernels-alone pass.
# usage: r3_ab.sh <tag> "<env A>" "<env B>" [more...]     (env "-" = no switch)
tag=$1; shift
i=0
for e in "$@"; do
  i=$((i+1))
  ee="$e"; [ "$e" = "-" ] && ee="OCTSEG_NOP=1"
  env $ee OCTSEG_PROFILE_DUMP=gpurun_out/${tag}_layers$i.csv python bench.py --steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/${tag}_ab$i.json 2> gpurun_out/${tag}_ab$i.err || { tail -5 gpurun_out/${tag}_ab$i.err; exit 1; }
  echo "== [$e]"; python - <<PY
import json
d = json.load(open('gpurun_out/${tag}_ab$i.json')); r = d['roofline']
print('frames/s', d['value'], 'ms/step', d['ms_per_step'], 'mfma alone', r['kernel_ms_per_step'], {k: v['ms_per_step'] for k, v in r['by_class'].items()}, 'hbm', d.get('roofline_hbm', {}).get('kernel_ms_per_step'))
PY
  python tools/group_layers.py gpurun_out/${tag}_layers$i.csv 2 | grep -v "^bn sweeps" | head -24
done

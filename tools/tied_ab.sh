#!/bin/bash
# A/B of the tied decomposition on a bench workload: per pass and together (one summary line each).  usage: tied_ab.sh "<modes>" [bench args]
modes=${1:-"- w dw fdw"}; shift
out=gpurun_out/tied_ab.txt; : > $out
for t in $modes; do
  [ "$t" = "-" ] && t="0"
  echo "== OCTSEG_TIED=$t $*" >> $out
  OCTSEG_TIED=$t python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print(o['value'], 'frames/s', o['ms_per_step'], 'ms/step  mfma', r['kernel_ms_per_step'], 'ms', r['achieved'], 'TF/s', {k: v['ms_per_step'] for k, v in r['by_class'].items()}, r.get('executed'))
" >> $out || exit 1
done
cat $out

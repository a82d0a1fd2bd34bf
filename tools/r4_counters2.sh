#!/bin/bash
# GPU box (under gpurun): SQ + TCC counters of the fused ConvTranspose2d weight gradient (wgrad_convt16_kernel) and, beside it, of the
# per-parity window-row-major launches it replaces (OCTSEG_NO_WGRAD_CONVT16) -> gpurun_out/r4_sq_counters_convt16_*.txt
set -e
cd /root/repo; export TMPDIR=/tmp; export TR=1
pmc() {  # tag, bench_conv.py args
  tag=$1; shift
  bash tools/pmc_shape.sh $tag "$@" > gpurun_out/r4_sq_counters_$tag.txt 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_${tag}_$ctr -- python3 tools/bench_conv.py "$@" > /dev/null 2>&1 || true
    python3 - <<PY >> gpurun_out/r4_sq_counters_$tag.txt
import csv,glob
fs=glob.glob('gpurun_out/pmc_${tag}_${ctr}/*/*counter_collection.csv')
if fs:
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if 'wgrad' in r['Kernel_Name']]
    f = 2.0 if '${ctr}'=='FETCH_SIZE' else 1.0
    if v: print('${tag} ${ctr} bytes per launch (KiB->B, FETCH x2 gfx950):', f*1024*sum(v)/len(v), len(v), 'launches')
PY
    rm -rf gpurun_out/pmc_${tag}_$ctr
  done
  python3 tools/bench_conv.py "$@" >> gpurun_out/r4_sq_counters_$tag.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_[1-4] gpurun_out/pmc_${tag}_[1-4].log
  echo "done $tag"
}
pmc convt16_512_128_176 16 176 176 512 128 4 2 wgrad
export OCTSEG_NO_WGRAD_CONVT16=1
pmc convt4x4_512_128_176 16 176 176 512 128 4 2 wgrad

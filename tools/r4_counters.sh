#!/bin/bash
# GPU box (under gpurun): round-4 counter evidence.
#  (a) SQ + TCC counters of wgrad_mfma_kernel<bf16,9> on x_1_2.conv1 / x_1_1.conv1 / x_2_2.conv1 and of <bf16,1> on a layer3 bottleneck
#      shape, of thin_conv_kernel, the affine-free conv3x3p data gradient and the gemm1x1 weight gradient -> gpurun_out/r4_sq_counters_*.txt
#  (b) rocprofv3 kernel stats of the four f4 workloads -> gpurun_out/r4_kernel_stats_<workload>.csv
set -e
cd /root/repo; export TMPDIR=/tmp
pmc() {  # tag, bench_conv.py args
  tag=$1; shift
  bash tools/pmc_shape.sh $tag "$@" > gpurun_out/r4_sq_counters_$tag.txt 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_${tag}_$ctr -- python3 tools/bench_conv.py "$@" > /dev/null 2>&1 || true
    python3 - <<PY >> gpurun_out/r4_sq_counters_$tag.txt
import csv,glob
fs=glob.glob('gpurun_out/pmc_${tag}_${ctr}/*/*counter_collection.csv')
if fs:
    v=[float(r['Counter_Value']) for r in csv.DictReader(open(fs[0])) if any(k in r['Kernel_Name'] for k in ('conv_mfma','conv3x3p','gemm1x1','wgrad','thin_'))]
    f = 2.0 if '${ctr}'=='FETCH_SIZE' else 1.0
    if v: print('${tag} ${ctr} bytes per launch (KiB->B, FETCH x2 gfx950):', f*1024*sum(v)/len(v), len(v), 'launches')
PY
    rm -rf gpurun_out/pmc_${tag}_$ctr
  done
  python3 tools/bench_conv.py "$@" >> gpurun_out/r4_sq_counters_$tag.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_[1-4] gpurun_out/pmc_${tag}_[1-4].log
  echo "done $tag"
}
pmc wgrad9_x12c1 16 176 176 1024 256 3 1 wgrad
pmc wgrad9_x11c1 16 88 88 1536 512 3 1 wgrad
pmc wgrad9_x22c1 16 176 176 768 256 3 1 wgrad
pmc wgrad1_l3c1 16 44 44 1024 256 1 1 wgrad
pmc wgrad1_l3c3 16 44 44 256 1024 1 1 wgrad
pmc thin_x04c1 16 704 704 32 16 3 1 fwd
pmc p3_dgrad_x22 16 176 176 768 256 3 1 dgrad
pmc g1_fwd_l3c3 16 44 44 256 1024 1 1 fwd
for w in fpn_r50_704 deeplabv3plus_r50_704 pspnet_r50_704 deeplabv3_r50_704; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r4_$w -- python3 bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r4_$w.log 2>&1
  python3 tools/prof_summary.py $(ls gpurun_out/prof_r4_$w/*/*.db | head -1) gpurun_out/r4_kernel_stats_$w.csv 6 > gpurun_out/r4_kernel_stats_$w.txt
  rm -rf gpurun_out/prof_r4_$w
  echo "done $w"
done

#!/bin/bash
# GPU box: SQ counter passes for one conv shape (bench_conv.py arguments).  usage: pmc_shape.sh <tag> N H W Cin Cout R [stride mode]
set -e
tag=$1; shift
cd /root/repo; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}_$i -- python3 tools/bench_conv.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1 || true
done
python3 - <<PY
import csv,glob,collections
for i in range(1,5):
    fs=glob.glob('gpurun_out/pmc_${tag}_%d/*/*counter_collection.csv'%i)
    if not fs: print('pass',i,'no output'); continue
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(fs[0])):
        if any(k in r['Kernel_Name'] for k in ('conv_mfma', 'conv3x3p', 'gemm1x1', 'wgrad', 'thin_')):
            a=agg[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    for k,v in agg.items(): print('${tag}',k, v[0]/v[1], 'per launch,', v[1], 'launches')
PY

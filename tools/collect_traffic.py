"""GPU box: HBM traffic of the dominant kernel from the TCC counters, as MI355X_MICROARCH.md prescribes
(separate --pmc passes for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled on gfx950 for wide coalesced reads;
both are in KiB).  Writes profiles/<tag>_traffic.json, which bench.py reports as roofline.traffic.
usage: collect_traffic.py <tag>      (run from the repo root, under gpurun; NOT under rocprofv3 itself)"""
import csv, glob, json, os, subprocess, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
os.environ['TMPDIR'] = '/tmp'
res = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    out = f'gpurun_out/pmc_{tag}_{ctr}'
    subprocess.run(['rocprofv3', '--pmc', ctr, '--output-format', 'csv', '-d', out, '--', 'python3', 'bench.py', '--steps', '1',
                    '--warmup', '1', '--no-cpu-baseline'], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(f'{out}/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
        agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    res[ctr] = {k: (v[0], v[1]) for k, v in agg.items()}
rows = {}
for k in res['FETCH_SIZE']:
    f_kib, n = res['FETCH_SIZE'][k]
    w_kib, _ = res['WRITE_SIZE'].get(k, (0.0, n))
    rows[k] = {'launches': n, 'fetch_bytes_per_launch': 2.0 * f_kib * 1024 / n, 'write_bytes_per_launch': w_kib * 1024 / n}
FAMILY = ('conv_mfma_kernel', 'conv3x3p_kernel', 'wgrad_mfma_kernel', 'wgrad1x1_kernel', 'wgrad_convt16_kernel', 'gemm1x1_kernel', 'thin_conv_kernel', 'thin_wgrad_kernel')
conv = {k: v for k, v in rows.items() if any(f in k for f in FAMILY)}
tot_l = sum(v['launches'] for v in conv.values())
summary = {'kernels': rows,
           'mfma_family': {'launches': tot_l,
                           'hbm_bytes_per_launch': sum((v['fetch_bytes_per_launch'] + v['write_bytes_per_launch']) * v['launches'] for v in conv.values()) / max(1, tot_l)},
           'note': 'FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes; 5 steps (1 warm-up + 1 timed + 3 of the kernels-alone pass) of bench.py, U-Net++/resnet101 704^2 bf16 B=16'}
summary['mfma_family']['hbm_bytes_per_step'] = summary['mfma_family']['hbm_bytes_per_launch'] * tot_l / 5.0
os.makedirs('profiles', exist_ok=True)
json.dump(summary, open(f'profiles/{tag}_traffic.json', 'w'), indent=1)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(summary, open(f'gpurun_out/{tag}_traffic.json', 'w'), indent=1)
print(json.dumps(summary['mfma_family']))

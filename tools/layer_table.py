"""Per-layer fwd / dgrad / wgrad table (ms per step, algorithmic TFLOP/s) from an OCTSEG_PROFILE_DUMP csv.
usage: layer_table.py file.csv [prefix]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pre = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.OrderedDict()
for r in rows:
    if r['class'] == 'hbm' or not r['layer'].startswith(pre):
        continue
    a = agg.setdefault((r['layer'], r['class']), [0.0, 0.0, 0])
    a[0] += float(r['ms']); a[1] += float(r['gflop']); a[2] += 1
nsteps = min(a[2] for a in agg.values())
lay = collections.OrderedDict()
for (l, c), a in agg.items():
    lay.setdefault(l, {})[c] = (a[0] / nsteps, a[1] / nsteps)
print(f'{"layer":40s} {"GF":>7s} | fwd ms  TF/s | dgrad ms TF/s | wgrad ms TF/s')
tot = collections.Counter()
for l, d in lay.items():
    gf = max(v[1] for v in d.values())
    s = f'{l:40s} {gf:7.1f} |'
    for c in ('fwd', 'dgrad', 'wgrad'):
        if c in d:
            ms, g = d[c]; s += f' {ms:6.3f} {g / ms if ms else 0:5.0f} |'; tot[c] += ms
        else:
            s += '              |'
    print(s)
print('totals', dict(tot))

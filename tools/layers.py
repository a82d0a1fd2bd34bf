import csv, collections, sys
def load(p):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        a = agg.setdefault((r['layer'], r['class']), [0.0, 0.0]); a[0] += float(r['ms']); a[1] += float(r['gflop'])
    return agg
a = load(sys.argv[1]); b = load(sys.argv[2]) if len(sys.argv) > 2 else None
n = 3 if len(sys.argv) <= 3 else int(sys.argv[3])
items = sorted(a.items(), key=lambda kv: -kv[1][0])
print('total ms/step', sum(v[0] for v in a.values()) / n)
for (l, c), (ms, gf) in items[:40]:
    extra = ''
    if b and (l, c) in b:
        extra = f'   was {b[(l,c)][0]:.3f} ms'
    print(f'{l:40s} {c:6s} {ms/n:8.3f} ms {gf/n:9.1f} GF {gf/ms:8.1f} TF/s{extra}')

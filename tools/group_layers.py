"""Per-group summary of an OCTSEG_PROFILE_DUMP csv (kernels alone, octseg_debug_set_serial pass).  usage: group_layers.py file.csv nsteps"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def group(layer, cls):
    if cls == 'hbm':
        return 'bn sweeps ' + layer.split('.')[-1]
    if layer.startswith('decoder') or layer.startswith('segmentation_head'):
        m = re.search(r'x_0_4|segmentation_head', layer)
        return 'decoder thin (x_0_4, head)' if m else 'decoder 3x3 / 1x1'
    if layer == 'encoder.conv1':
        return 'stem'
    if re.search(r'conv[13]$|downsample\.0$', layer) and re.search(r'layer\d\.\d+\.conv3|layer\d\.\d+\.downsample', layer):
        return 'bottleneck 1x1'
    if re.search(r'conv1$', layer):
        return 'bottleneck 1x1 / basic 3x3 conv1'
    return 'encoder 3x3'


agg = collections.OrderedDict()
for r in rows:
    k = (group(r['layer'], r['class']), r['class'])
    a = agg.setdefault(k, [0.0, 0.0, 0])
    a[0] += float(r['ms']); a[1] += float(r['gflop']); a[2] += 1
print(f"{'group':40s} {'class':6s} {'ms/step':>9s} {'GF|GB':>10s} {'TF/s|TB/s':>10s} {'launches':>8s}")
for (g, c), (ms, gf, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f'{g:40s} {c:6s} {ms / n:9.3f} {gf / n:10.1f} {gf / ms:10.1f} {cnt / n:8.0f}')
tot = collections.defaultdict(float)
for (g, c), (ms, gf, cnt) in agg.items():
    tot[c] += ms / n
print('totals ms/step:', dict(tot))

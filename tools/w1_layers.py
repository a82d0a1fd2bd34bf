"""Per-layer table of the bottleneck 1x1 weight gradients from OCTSEG_PROFILE_DUMP csv files: python tools/w1_layers.py a.csv b.csv ..."""
import collections, csv, sys
tabs = []
for f in sys.argv[1:]:
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['class'] == 'wgrad' and r['layer'].startswith('encoder.layer') and any(k in r['layer'] for k in ('conv1', 'conv3', 'downsample')):
            d[r['layer']].append(float(r['ms']))
    tabs.append({k: sum(v) / len(v) for k, v in d.items()})
keys = [k for k in tabs[0] if any(s in k for s in ('layer1.0', 'layer1.1', 'layer2.0', 'layer2.1', 'layer3.0', 'layer3.1.', 'layer4.0', 'layer4.1'))]
for k in keys:
    print(f'{k:36s}' + ''.join(f'{t.get(k, 0) * 1000:9.1f}' for t in tabs) + ' us')
print(f'{"sum of all bottleneck 1x1 wgrads":36s}' + ''.join(f'{sum(t.values()):9.3f}' for t in tabs) + ' ms')

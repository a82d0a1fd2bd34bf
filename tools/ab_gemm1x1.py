"""GPU box: the benchmark step's gradient arena with and without gemm1x1.hip (OCTSEG_NO_GEMM1X1=1), per-parameter comparison.
usage: ab_gemm1x1.py [arch enc classes B S]"""
import os, subprocess, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    from oct_segmentation_amd.engine import SegNet
    from synth import make_batch
    arch, enc, classes, B, S, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    net = SegNet(arch, enc, classes=classes, device='cuda', compute_dtype=torch.bfloat16, seed=3).train()
    img, mask = make_batch(B, classes, S, seed=21, empty_last=(classes > 1))
    loss, logits, stats = net.train_step_raw(img.cuda(), mask.cuda(), normalize=True, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    torch.cuda.synchronize()
    torch.save({'buf': net.bn_buffers.cpu(), 'bn': net.bn_table, 'g': net._grad_arena.cpu(), 'logits': logits.cpu(), 'loss': loss.item(), 'table': net.param_table}, out)
    sys.exit(0)
import torch
cfg = sys.argv[1:6] if len(sys.argv) > 5 else ['unetplusplus', 'resnet101', '1', '16', '704']
res = {}
SW = os.environ.get('AB_SWITCH', 'OCTSEG_NO_GEMM1X1')
for tag, env in (('new', {}), ('old', {SW: '1'})):
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, __file__, 'child'] + cfg + [f'/tmp/ab_{tag}.pt'], check=True, env=e)
    res[tag] = torch.load(f'/tmp/ab_{tag}.pt', weights_only=False)
a, b = res['new'], res['old']
print('loss', a['loss'], b['loss'], 'logits max|d|', (a['logits'] - b['logits']).abs().max().item(), 'scale', b['logits'].abs().max().item())
d = (a['logits'] - b['logits']).abs()
print('logits |d| mean', d.mean().item(), 'p99', d.flatten().kthvalue(int(0.99 * d.numel())).values.item())
worst = []
for bn in a['bn']:
    for key in ('mean_offset', 'var_offset'):
        x, y = a['buf'][bn[key]:bn[key] + bn['C']], b['buf'][bn[key]:bn[key] + bn['C']]
        worst.append(((x - y).abs().max().item() / max(y.abs().max().item(), 1e-6), bn['name'], key))
worst.sort(reverse=True)
print('BN running statistics, worst relative differences:', worst[:8])
ga, gb = a['g'].double(), b['g'].double()
print('global cosine', float((ga * gb).sum() / (ga.norm() * gb.norm())), 'nonzero fraction new / old', float((ga != 0).float().mean()), float((gb != 0).float().mean()))
rows = []
for p in a['table']:
    x, y = ga[p['offset']:p['offset'] + p['numel']], gb[p['offset']:p['offset'] + p['numel']]
    d = (x - y).abs().max().item() / max(y.abs().max().item(), 1e-30)
    rows.append((d, p['name'], float((x != 0).float().mean()), float((y != 0).float().mean())))
rows.sort(reverse=True)
for r in rows[:15]:
    print(f'{r[1]:45s} rel diff {r[0]:.3e}  nonzero new {r[2]:.3f} old {r[3]:.3f}')

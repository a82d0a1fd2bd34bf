"""GPU box: how much of the timed steps is covered by at least one kernel (union of the kernel intervals of a rocprofv3 --kernel-trace .db),
and by the MFMA kernels alone.  usage: gpu_busy.py <db> <first_step> <n_steps>   -- steps are delimited by the optimizer kernels (bench.py: warm-up, timed, then 3 one-stream steps)"""
import sqlite3, sys
db = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
opt = [r for r in rows if 'optim_kernel' in r[0]]
if len(opt) < first + nlast:
    print('not enough steps in the trace', len(opt)); sys.exit(1)
t0, t1 = opt[first - 1][2], opt[first - 1 + nlast][2]          # end of the optimizer kernel of step first - 1 .. of step first - 1 + n
win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
busy = union([(r[1], r[2]) for r in win])
mf = [(r[1], r[2]) for r in win if any(k in r[0] for k in ('conv_mfma', 'conv3x3p', 'gemm1x1', 'wgrad_mfma', 'thin_conv', 'thin_wgrad'))]
gaps = []
import collections, re
pair = collections.Counter(); pairn = collections.Counter()
short = lambda n: re.sub(r'<.*', '', n.replace('(anonymous namespace)::', '').replace('void ', ''))[:28]
iv = sorted((r[1], r[2], r[0]) for r in win); ce = iv[0][1]; last = iv[0][2]
for s, e, nm in iv[1:]:
    if s > ce:
        gaps.append(s - ce); pair[(short(last), short(nm))] += s - ce; pairn[(short(last), short(nm))] += 1
    if e > ce: ce = e; last = nm
print(f'{nlast} steps: wall {(t1 - t0) / 1e6 / nlast:.3f} ms/step, some kernel running {busy / 1e6 / nlast:.3f} ms/step ({100 * busy / (t1 - t0):.1f} %), '
      f'an MFMA kernel running {union(mf) / 1e6 / nlast:.3f} ms/step; {len(gaps) // nlast} idle gaps per step, '
      f'sum {sum(gaps) / 1e6 / nlast:.3f} ms, median {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us')
hist = collections.Counter(min(int(g / 2000) * 2, 40) for g in gaps)
print('gap histogram (us: count per step):', ' '.join(f'{k}-{k + 2}:{v // nlast}' for k, v in sorted(hist.items())))
print('idle time by (kernel that ended last -> kernel that starts), ms per step:')
for k, v in pair.most_common(14):
    print(f'  {k[0]:28s} -> {k[1]:28s} {v / 1e6 / nlast:7.3f} ms  {pairn[k] // nlast:4d} gaps  {v / pairn[k] / 1e3:5.1f} us each')
durs = collections.Counter(); cnt = collections.Counter()
for r in win: durs[short(r[0])] += r[2] - r[1]; cnt[short(r[0])] += 1
print('kernel time per step (overlapped durations):')
for k, v in durs.most_common(16): print(f'  {k:28s} {v / 1e6 / nlast:7.3f} ms {cnt[k] // nlast:5d} launches')

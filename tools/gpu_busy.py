"""GPU box: how much of the timed steps is covered by at least one kernel (union of the kernel intervals of a rocprofv3 --kernel-trace .db),
and by the MFMA kernels alone.  usage: gpu_busy.py <db> <first_step> <n_steps>   -- steps are delimited by the optimizer kernels (bench.py: warm-up, timed, then 3 one-stream steps)"""
import sqlite3, sys
db = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 3
MFMA = ('conv_mfma', 'conv3x3p', 'gemm1x1', 'wgrad_mfma', 'wgrad1x1', 'wgrad_convt16', 'thin_conv', 'thin_wgrad')
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
mf = [(r[1], r[2]) for r in win if any(k in r[0] for k in MFMA)]
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

# exposed time of the other kernels: the part of each interval during which no MFMA kernel runs (what overlap does NOT hide)
mu = []
for s_, e_ in sorted(mf):
    if mu and s_ <= mu[-1][1]: mu[-1][1] = max(mu[-1][1], e_)
    else: mu.append([s_, e_])
import bisect
starts = [m[0] for m in mu]
def exposed(s_, e_):
    cov = 0; i = max(0, bisect.bisect_right(starts, s_) - 1)
    while i < len(mu) and mu[i][0] < e_:
        cov += max(0, min(e_, mu[i][1]) - max(s_, mu[i][0])); i += 1
    return (e_ - s_) - cov
ex = collections.Counter(); tot = collections.Counter()
for r in win:
    if any(k in r[0] for k in MFMA): continue
    ex[short(r[0])] += exposed(r[1], r[2]); tot[short(r[0])] += r[2] - r[1]
print('non-MFMA kernels: duration and the part of it with NO MFMA kernel running beside (ms per step):')
for k, v in ex.most_common(12): print(f'  {k:28s} {tot[k] / 1e6 / nlast:7.3f} ms, exposed {v / 1e6 / nlast:7.3f}')
print(f'  total exposed {sum(ex.values()) / 1e6 / nlast:.3f} ms per step')

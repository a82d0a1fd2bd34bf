"""Summarise a rocprofv3 results .db (kernel-trace) into a per-kernel CSV + table: python tools/prof_summary.py <db> [out.csv] [steps]"""
import csv, re, sqlite3, sys

db = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
if out:
    with open(out, 'w') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 2), r[4], r[5]])
print(f'total kernel time {tot / 1e6:.2f} ms over {steps} steps')
for r in rows[:30]:
    n = re.sub(r'\(anonymous namespace\)::', '', r[0])
    n = re.sub(r'\(.*', '', n)[:78]
    print(f'{n:78s} {r[1]:6d} {r[2] / 1e6 / steps:8.2f} ms/step {100 * r[2] / tot:5.1f}%')

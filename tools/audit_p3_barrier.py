"""Build-time check of conv3x3p_kernel's interleaved tap (csrc/conv3x3p.hip, ILV): the tap ends with `s_waitcnt lgkmcnt(N)` + `s_barrier`,
which is only right if the N youngest LGKM operations in front of it are fragment reads (the last slots of the tap) -- LDS operations retire in
order, so everything older (this wave's LDS stores of slots 0, 1 and 6) has then landed.  hipcc does not model that wait: a compiler or flag
change that moves an LDS store (or any scalar memory load, which retires out of order in the same counter) into the last four would be a
silent race.  usage: audit_p3_barrier.py conv3x3p.s     (exit 1 on a violation or when no such barrier is found)"""
import re
import sys

src = open(sys.argv[1]).read().split('\n')
checked = bad = 0
for i, line in enumerate(src):
    m = re.match(r'\s*s_waitcnt lgkmcnt\(([1-9])\)\s*$', line.split(';')[0])
    if not m:
        continue
    nkeep = int(m.group(1))
    j = i + 1
    while j < len(src) and (not src[j].strip() or src[j].strip().startswith(';')):
        j += 1
    if j >= len(src) or not src[j].strip().startswith('s_barrier'):
        continue
    checked += 1
    young = []
    k = i - 1
    while k >= 0 and len(young) < nkeep:
        ins = src[k].split(';')[0].strip()
        if ins.endswith(':') or ins.startswith('s_cbranch') or ins.startswith('s_branch') or ins.startswith('s_barrier'):
            break                                   # left the tap's straight-line code: fewer than four operations in it
        if ins.startswith('ds_') or ins.startswith('s_load') or ins.startswith('s_buffer_load') or ins.startswith('s_memtime'):
            young.append(ins)
        k -= 1
    if len(young) < nkeep or not all(x.startswith('ds_read_b128') for x in young):
        bad += 1
        print(f'line {i + 1}: the {nkeep} youngest LGKM operations before the barrier are {young}', file=sys.stderr)
print(f'{checked} interleaved-tap barriers audited, {bad} violations')
sys.exit(1 if bad or checked == 0 else 0)

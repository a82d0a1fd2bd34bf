"""ISA audit of conv_mfma.hip (CPU, needs hipcc): the window slices of run9r are loaded by asm global_load
statements that hipcc does not track (and so are the lazy-BN parameters of a chunk).  Between such a load and the
`s_waitcnt vmcnt(N)` that retires it (the first one with at least N memory operations issued behind the load) NO
instruction may read or write the destination registers -- a compiler copy there reads registers still in flight.
usage: python tools/audit_asm_loads.py   (exit code 1 on a violation)"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'oct_segmentation_amd', 'csrc', 'conv_mfma.hip')
out = os.path.join(tempfile.mkdtemp(), 'conv.s')
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-w', '-S', '--cuda-device-only', '-o', out, src], check=True)
s = open(out).read()
bad = nload = 0
for nm in re.findall(r'^(_ZN6octseg16conv_mfma_kernelI\S+):', s, re.M):
    i = s.index(nm + ':'); j = s.index('.Lfunc_end', i)
    lines = s[i:j].split('\n')
    # basic-block successors are not followed: the unrolled chunk body is straight-line apart from exec-masked
    # branches that rejoin, so a linear scan to the second asm wait covers every path
    inasm = False
    loads = []
    for k, l in enumerate(lines):
        t = l.strip()
        if 'ASMSTART' in t: inasm = True; continue
        if 'ASMEND' in t: inasm = False; continue
        if inasm and t.startswith('global_load_dwordx4'):
            m = re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\]', t)
            loads.append((k, int(m.group(1)), int(m.group(2))))
    for k, lo, hi in loads:
        nload += 1
        # vmcnt retires in issue order: `s_waitcnt vmcnt(N)` leaves at most the N youngest memory operations in flight, so
        # a load is complete at the first wait that has at least N operations issued behind the load (M >= N)
        younger = 0
        for q in range(k + 1, len(lines)):
            t = lines[q].strip()
            if 'ASMSTART' in t or 'ASMEND' in t or t.startswith(';') or not t: continue
            mw = re.match(r's_waitcnt .*vmcnt\((\d+)\)', t)
            if mw:
                if younger >= int(mw.group(1)): break
                continue
            if re.match(r'(global|buffer|scratch|flat)_', t): younger += 1
            for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', t):
                a, b = (int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)), int(m.group(3)))
                if not (b < lo or a > hi):
                    print(f'{nm[29:52]}: load @{k} v[{lo}:{hi}] touched @{q}: {t[:70]}'); bad += 1
print(f'{nload} asm loads audited, {bad} violations')
sys.exit(1 if bad else 0)

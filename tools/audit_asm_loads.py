"""ISA audit of conv_mfma.hip (CPU, needs hipcc): the window slices of run9r are loaded by asm global_load
statements that hipcc does not track (and so are the lazy-BN parameters of a chunk).  Between such a load and the
`s_waitcnt vmcnt(N)` that retires it (the first one with at least N memory operations issued behind the load) NO
instruction may read or write the destination registers -- a compiler copy there reads registers still in flight.
The counted waits assume vmcnt retires in issue order.  That holds for global / buffer / scratch operations (segment-specific
encodings); a generic FLAT operation in flight makes the counter out of order (LLVM SIInsertWaitcnts: hasPendingFlat), so a
flat_ instruction that can execute while an asm load is in flight is a violation: with one pending only vmcnt(0) proves
anything.  (Register spills to scratch in a prologue / epilogue, where no asm load is in flight, are legal: the walk below
only looks at what can run between a load and its wait.)
Part of the build: `make -C oct_segmentation_amd/csrc` (hence __graft_entry__.build()) runs it on the ISA of the very
flags it compiles the library with and fails on a violation or when it finds no asm loads to audit.
usage: python tools/audit_asm_loads.py [file.s]   (exit code 1 on a violation; without an argument the ISA is generated here)"""
import os, re, subprocess, sys, tempfile
VALIDATED_WITH = 'HIP version: 7.2.26015'   # compiler the shipped schedule was last hand-checked with (informational: the audit
                                            # itself runs on whatever compiler builds the library)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'oct_segmentation_amd', 'csrc', 'conv_mfma.hip')
hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
if len(sys.argv) > 1:
    out = sys.argv[1]
else:
    out = os.path.join(tempfile.mkdtemp(), 'conv.s')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-w', '-S', '--cuda-device-only', '-o', out, src], check=True)
try:
    ver = subprocess.run([hipcc, '--version'], capture_output=True, text=True).stdout.splitlines()[0]
    if not ver.startswith(VALIDATED_WITH):
        print(f'note: compiler "{ver}" differs from the one the schedule was hand-checked with ("{VALIDATED_WITH}"); auditing its ISA')
except Exception:
    pass
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
    labels = {l.strip().split(':')[0]: idx for idx, l in enumerate(lines) if re.match(r'\s*\.LBB\d+_\d+:', l)}
    for k, lo, hi in loads:
        nload += 1
        # vmcnt retires in issue order: `s_waitcnt vmcnt(N)` leaves at most the N youngest memory operations in flight, so
        # a load is complete at the first wait that has at least N operations issued behind the load (M >= N).
        # Every control-flow path from the load is walked (branch targets and fall-throughs) until such a wait.
        seen = set(); flagged = set()
        work = [(k + 1, 0)]
        while work:
            q, younger = work.pop()
            while q < len(lines):
                if (q, younger) in seen: break
                seen.add((q, younger))
                t = lines[q].strip(); q += 1
                if q - 1 == k: break                                   # issued again: a new flight
                if 'ASMSTART' in t or 'ASMEND' in t or t.startswith(';') or not t or t.startswith('.'): continue
                mw = re.match(r's_waitcnt .*vmcnt\((\d+)\)', t)
                if mw:
                    if younger >= int(mw.group(1)): break
                    continue
                if t.startswith('s_endpgm'): break
                mb = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', t)
                if mb:
                    if mb.group(1) in labels: work.append((labels[mb.group(1)], younger))
                    continue
                mb = re.match(r's_branch (\.LBB\d+_\d+)', t)
                if mb:
                    q = labels[mb.group(1)]
                    continue
                if re.match(r'flat_', t) and (q - 1) not in flagged:
                    flagged.add(q - 1)
                    print(f'{nm[29:52]}: load @{k} in flight across {t.split()[0]} @{q - 1}: out-of-order vmcnt'); bad += 1
                if re.match(r'(global|buffer|scratch)_', t): younger = min(younger + 1, 64)   # in-order vmcnt operations
                for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', t):
                    a, b = (int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)), int(m.group(3)))
                    if not (b < lo or a > hi) and (q - 1) not in flagged:
                        flagged.add(q - 1)
                        print(f'{nm[29:52]}: load @{k} v[{lo}:{hi}] touched @{q - 1}: {t[:70]}'); bad += 1
print(f'{nload} asm loads audited, {bad} violations')
if nload == 0:
    print('no asm loads found: the kernels under audit were renamed or the ISA is not conv_mfma.hip -- fix the audit, do not skip it')
sys.exit(1 if (bad or nload == 0) else 0)

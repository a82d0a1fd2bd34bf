"""CPU check of the conv kernels' ISA (cross-compiled for gfx950, no GPU needed): the window slices and lazy-BN
parameters of the unrolled main loops are fetched by asm loads that hipcc's waitcnt pass does not track; between
such a load and the counted `s_waitcnt vmcnt(N)` that retires it no instruction may touch its destination registers
(tools/audit_asm_loads.py walks every control-flow path).  A violation is a data race on real hardware that the parity
tests may or may not catch, so it is caught here."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_no_instruction_touches_an_asm_load_destination_in_flight():
    # the build (make -C oct_segmentation_amd/csrc, i.e. __graft_entry__.build()) leaves the ISA it audited next to the objects: reuse it
    # when it is newer than every source it was generated from, regenerate it otherwise (4 minutes of hipcc)
    csrc = os.path.join(ROOT, 'oct_segmentation_amd', 'csrc')
    isa = os.path.join(csrc, 'build', 'conv_mfma.s')
    deps = [os.path.join(csrc, f) for f in ('conv_mfma.hip', 'common.h', 'conv_common.h', 'kernels.h')]
    args = [isa] if os.path.exists(isa) and all(os.path.getmtime(isa) >= os.path.getmtime(d) for d in deps) else []
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'audit_asm_loads.py')] + args, capture_output=True, text=True, timeout=1500)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0, tail
    n = int(tail.split()[0])
    assert n >= 100 and tail.endswith('0 violations'), tail   # the loops under audit exist (run9r, run1p) and are clean


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_conv3x3p_barrier_leaves_only_fragment_reads_in_flight(tmp_path):
    """conv3x3p's interleaved tap ends with `s_waitcnt lgkmcnt(N)` + `s_barrier` (hipcc does not model it): the N youngest LGKM operations
    must be fragment reads in each of the 18 unrolled taps of both interleaved instantiations (with / without the staging affine; tools/audit_p3_barrier.py); and the checker itself must
    reject an LDS store moved into that window."""
    csrc = os.path.join(ROOT, 'oct_segmentation_amd', 'csrc')
    isa = os.path.join(csrc, 'build', 'conv3x3p.s')
    deps = [os.path.join(csrc, f) for f in ('conv3x3p.hip', 'common.h', 'conv_common.h', 'kernels.h')]
    if not (os.path.exists(isa) and all(os.path.getmtime(isa) >= os.path.getmtime(d) for d in deps)):
        isa = str(tmp_path / 'conv3x3p.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-w', '-S', '--cuda-device-only',
                        os.path.join(csrc, 'conv3x3p.hip'), '-o', isa], check=True, timeout=900)
    tool = os.path.join(ROOT, 'tools', 'audit_p3_barrier.py')
    r = subprocess.run([sys.executable, tool, isa], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith('36 interleaved-tap barriers audited, 0 violations'), r.stdout
    # negative control: swap the youngest fragment read of the first audited barrier for an LDS store
    lines = open(isa).read().split('\n')
    def next_ins(i):
        i += 1
        while not lines[i].strip() or lines[i].strip().startswith(';'):
            i += 1
        return lines[i].strip()
    import re
    k = next(i for i, l in enumerate(lines) if re.match(r's_waitcnt lgkmcnt\([1-9]\)$', l.strip()) and next_ins(i).startswith('s_barrier'))
    j = next(i for i in range(k - 1, 0, -1) if lines[i].strip().startswith('ds_read_b128'))
    lines[j] = '\tds_write_b128 v0, v[0:3]'
    broken = tmp_path / 'broken.s'
    broken.write_text('\n'.join(lines))
    r = subprocess.run([sys.executable, tool, str(broken)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and '1 violations' in r.stdout, r.stdout + r.stderr

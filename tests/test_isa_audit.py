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

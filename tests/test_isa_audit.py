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
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'audit_asm_loads.py')], capture_output=True, text=True, timeout=900)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0, tail
    n = int(tail.split()[0])
    assert n >= 100 and tail.endswith('0 violations'), tail   # the loops under audit exist (run9r, run1p) and are clean

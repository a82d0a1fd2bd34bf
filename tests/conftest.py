import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# conv_mfma.hip takes its 11 x 11 pixel tiles (3x3 layers on maps that are multiples of 11 and not of 16) only on grids that give every CU a
# workgroup -- a speed rule (OCTSEG_TILE11_MINWG, default 256, read once per process).  The parity tests run two or three frames: without this
# they would compare the 16-pixel tilings only.  Set before the library is first used; the 704^2 batch-16 tests cover the default routing.
os.environ.setdefault('OCTSEG_TILE11_MINWG', '1')


def _usable_cores():
    """CPU threads this process may really use (bench.py host_cores): the affinity mask, cut to the cgroup quota; a one-GPU job on the
    GPU boxes sees every core of the host but is scheduled on a 16-core share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, p = f.read().split()
            if q != 'max':
                return max(1, min(n, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return 16 if n > 32 else n


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle (torch / oneDNN) with one thread per VISIBLE core oversubscribes the box's share by an order of magnitude
    import torch
    torch.set_num_threads(_usable_cores())


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('GPU test selected but no GPU is visible')
    return torch.device('cuda:0')

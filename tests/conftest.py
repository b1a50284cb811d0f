import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """A fresh checkout has no libvittf.so (built artefacts are git-ignored): build it once (hipcc cross-compiles gfx950
    without a GPU) so that the C-ABI surface tests and every -m gpu test find the in-tree library."""
    lib = os.path.join(ROOT, 'vit-tf_amd', 'libvittf.so')
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(['make', '-C', os.path.join(ROOT, 'vit-tf_amd', 'csrc'), '-j', str(min(8, os.cpu_count() or 2))],
                       check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope='session', autouse=True)
def oracle_threads():
    """ONE place that sizes torch's CPU thread pool for the oracle: the cores this process may really use (affinity mask
    capped by the cgroup quota, at most 16 -- bench.host_cores).  A GPU box reports 256 CPUs to os.cpu_count() for a 16-CPU
    share, and a pool of that size makes every CPU reference in the session several times slower (GPUTEST_r03: the suite
    hit the driver's 900 s limit).  No test sets the thread count itself."""
    import torch
    from bench import host_cores
    before = torch.get_num_threads()
    torch.set_num_threads(host_cores())
    yield host_cores()
    torch.set_num_threads(before)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def gpu():
    """Skip-free guard for -m gpu tests: they must run on a GPU box with the library built."""
    import torch
    import vit_tf_amd as vt
    assert torch.cuda.is_available(), 'gpu-marked test started without a GPU'
    lib = vt._lib.load()
    assert lib.vittf_device_count() >= 1, 'no gfx950 device'
    return torch.device('cuda', 0)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLD


@pytest.fixture(scope='session')
def eld_lib():
    """The HIP library must be the thing under test on a GPU box: no fallback, fail loudly."""
    import eld_amd
    return eld_amd.load_library()


@pytest.fixture(autouse=True)
def _restore_torch_threads():
    """Some tests raise torch's CPU thread count for the oracle; reductions (e.g. the weight checksums compared bit-for-bit
    with tests/golden/unet.npz) depend on it, so every test starts from the same setting."""
    try:
        import torch
    except Exception:
        yield
        return
    n = torch.get_num_threads()
    yield
    if torch.get_num_threads() != n:
        torch.set_num_threads(n)

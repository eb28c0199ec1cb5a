import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# The oracle side of the GPU suite is torch on the HOST (fp64 autograd, double backward). On the GPU boxes (256 hardware
# threads) torch's default of 128 intra-op threads makes those small-tensor graphs 20 x slower than 8 - 16 threads do
# (tools/debug/oracle_threads.py: 10.7 s against 0.46 s for one double backward) -- most of the suite's wall time.
import os  # noqa: E402

import torch  # noqa: E402

torch.set_num_threads(max(1, min(12, os.cpu_count() or 1)))

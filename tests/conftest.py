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


# ---- the round-5/6 kernels inside the driver's suite ---------------------------------------------------
# By default the two-workgroups-per-CU row kernels (k_emlp_s, k_emlp_bwd_s, k_head_s, k_head_bwd_s, k_compress_bwd_s, k_comb_s,
# k_rowlin_s) serve graphs of at least 28 672 edges and the fused attention block graphs of at least 3 840 tiles: the small
# goldens of the three modules below would never reach them. Every test of those modules therefore runs twice: under the
# default policy and with those kernels forced on every graph (``emlp_s = 2, attn_fused = 7``), against the SAME goldens and
# the same 1e-5 bar.
FORCED_MODULES = {"test_gpu_parity", "test_gpu_variants", "test_gpu_backend"}
POLICIES = {"default": {"emlp_s": 1, "attn_fused": 3}, "forced": {"emlp_s": 2, "attn_fused": 7}}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.rsplit(".", 1)[-1] in FORCED_MODULES:
        metafunc.fixturenames.append("kernel_policy")
        metafunc.parametrize("kernel_policy", list(POLICIES), indirect=True)


@pytest.fixture
def kernel_policy(request):
    from metatrain_amd import runtime

    for k, v in POLICIES[request.param].items():
        runtime.config_set(k, v)
    yield request.param
    for k, v in POLICIES["default"].items():
        runtime.config_set(k, v)

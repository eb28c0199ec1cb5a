import sys
sys.path.insert(0, ".")
import torch
from metatrain_amd import runtime as rt
rt.config_set("emlp_s", 2)
rt.config_set("attn_fused", 7)
import pytest
sys.exit(pytest.main(["tests", "-q", "-m", "gpu", "-x", "--deselect", "tests/test_gpu_emlp_s.py", "-p", "no:cacheprovider"]))

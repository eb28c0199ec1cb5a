"""The bench scripts on a box without a GPU: they accept the driver's flags (`--gpus N --steps K --warmup W`) and then
stop with an error -- no JSON line, no CPU fallback (the oracle is only ever the `cpu_baseline` leg of a GPU run)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the point is the behaviour without a GPU")
@pytest.mark.parametrize("script", ["bench.py", "bench_train.py", "bench_soap.py", "bench_pet_box.py"])
def test_bench_refuses_to_run_without_a_gpu(script):
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert "needs MI355X GPUs" in r.stderr
    assert r.stdout.strip() == ""  # no metric line that could be mistaken for a measurement


def test_bench_flags_of_the_contract_are_declared():
    for script in ("bench.py", "bench_train.py", "bench_soap.py", "bench_pet_box.py"):
        text = open(os.path.join(ROOT, script)).read()
        for flag in ('"--gpus"', '"--steps"', '"--warmup"'):
            assert flag in text, (script, flag)

"""CPU tests of the TorchScript surface (csrc/torch_ops.cpp): the op library loads, the custom class scripts,
pickles through torch.jit.save / load with its weights, and refuses CPU tensors (no compute without a GPU)."""
import io
import os

import pytest
import torch

from metatrain_amd import build
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import synthetic_params


@pytest.fixture(scope="module")
def core():
    if not os.path.exists(build.TORCH_LIB):
        build.build(verbose=False)
        build.build_torch_ops(verbose=False)
    from metatrain_amd.pet import script

    hypers = default_hypers()
    params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
    return script.make_core(hypers, [1, 6, 7, 8], params, "energy"), len(params)


def test_scripts_and_round_trips_through_jit_save(core):
    from metatrain_amd.pet import script

    c, n_tensors = core
    assert c.num_tensors() == n_tensors
    mod = torch.jit.script(script.EnergyAndForces(c))
    buf = io.BytesIO()
    torch.jit.save(mod, buf)
    buf.seek(0)
    back = torch.jit.load(buf)
    assert back.pet.core.num_tensors() == n_tensors
    assert "atomic_energies" in str(back.pet.graph)


def test_exported_energy_model_scripts_with_optional_arguments(core):
    from metatrain_amd.pet import script

    comp = torch.zeros(9)
    comp[[1, 6, 7, 8]] = torch.tensor([-0.5, -37.8, -54.6, -75.1])
    mod = torch.jit.script(script.ExportedEnergyModel(core[0], 2.5, comp))
    buf = io.BytesIO()
    torch.jit.save(mod, buf)
    buf.seek(0)
    back = torch.jit.load(buf)
    assert float(back.scale) == 2.5 and torch.equal(back.composition, comp)
    assert "selected_atoms" in str(back.forward.schema)


def test_cpu_tensors_are_refused(core):
    from metatrain_amd.pet import script

    mod = script.PETScriptModule(core[0])
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU path"):
        mod(z(2, 3), z(1, 3, 3), z(0, dtype=torch.int32), z(0, dtype=torch.int32), z(0, 3, dtype=torch.int32),
            torch.tensor([1, 6]), z(2, dtype=torch.int32))


def test_backend_mirror_scripts_like_the_reference_module():
    """utils/testing/torchscript.py:39-75 at the backend level: ``torch.jit.script(PETBackend)`` compiles the three
    calls (each one a method of torch.classes.pet_hip.PetHipBackend), survives torch.jit.save / load with its
    parameters and a (target with several blocks and properties), and refuses CPU tensors."""
    from metatrain_amd.pet import PETBackend

    be = PETBackend(default_hypers(), [1, 6, 7, 8])
    be.add_output("energy", {"energy": [1]})
    be.add_output("multi", {"a": [3], "b": [3, 2]})
    n_params = len(list(be.parameters()))
    mod = torch.jit.script(be)
    for name in ("preprocess", "calculate_features", "predict"):
        assert hasattr(mod, name)
    assert "Dict(str, Tensor)" in str(mod.preprocess.schema) and "cutoff_width_adaptive" in str(mod.preprocess.schema)
    assert "requested_output_names" in str(mod.predict.schema)
    buf = io.BytesIO()
    torch.jit.save(mod, buf)
    buf.seek(0)
    back = torch.jit.load(buf)
    assert len(back._params()) == n_params
    assert [tuple(a.shape) for a in back._params()] == [tuple(a.shape) for a in be._params()]
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU path"):
        back.preprocess(z(2, 3), z(0, dtype=torch.int32), z(0, dtype=torch.int32), torch.tensor([1, 6]), z(1, 3, 3),
                        z(0, 3, dtype=torch.int32), z(2, dtype=torch.long), 1.0)

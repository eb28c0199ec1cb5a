"""CPU tests of the host-side PETBackend mirror: state-dict schema / init parity with the
reference (no GPU work: constructing the module owns parameters only)."""
import pytest
import torch

from metatrain_amd._lib import PetHipError
from metatrain_amd.pet import PETBackend, default_hypers
from metatrain_amd.synthetic import state_dict_schema
from oracle import pet as opet


def test_state_dict_keys_shapes_and_order_match_reference_schema():
    hypers = default_hypers()
    be = PETBackend(hypers, [1, 6, 7, 8])
    be.add_output("energy", {"energy": [1]})
    sd = be.state_dict()
    schema = state_dict_schema(hypers, [1, 6, 7, 8], {"energy": 1})
    assert list(sd.keys()) == [k for k, _, _ in schema]
    for k, shape, _ in schema:
        assert tuple(sd[k].shape) == tuple(shape), k
    # first entry is the integer buffer the reference's checkpoint dtype probe relies on
    assert next(iter(sd)) == "species_to_species_index" and sd["species_to_species_index"].dtype == torch.int64
    assert sum(p.numel() for p in be.parameters()) == 2903298  # default PET + energy head


def test_seed0_init_equals_reference_init():
    """torch.manual_seed(0) + the reference's construction order gives the reference's weights
    (the oracle's reference_init_params reproduces pet/tests/test_regression.py:66-74 with them)."""
    hypers = default_hypers()
    ref = opet.reference_init_params(hypers, [1, 6, 7, 8], "mtt::U0", seed=0)
    torch.manual_seed(0)
    be = PETBackend(hypers, [1, 6, 7, 8])
    be.add_output("mtt::U0", {"mtt::U0": [1]})
    sd = be.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


def test_reference_state_dict_loads_strictly():
    hypers = default_hypers()
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1})
    be = PETBackend(hypers, [1, 6, 7, 8])
    be.add_output("energy", {"energy": [1]})
    res = be.load_state_dict(params, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    be.remove_output("energy")
    assert not any(k.startswith(("node_heads", "edge_heads")) for k in be.state_dict())


VARIANTS = {
    "conditioned": dict(system_conditioning=True),
    "legacy": dict(normalization="LayerNorm", activation="SiLU", transformer_type="PostLN", featurizer_type="residual"),
    "layernorm": dict(normalization="LayerNorm"),
    "postln": dict(transformer_type="PostLN"),
    "residual": dict(featurizer_type="residual"),
}


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_variant_state_dicts_match_reference_schema_and_script(tag):
    """LayerNorm adds the norm biases, the residual featuriser drops the combination modules and has one node embedder,
    head and last layer per GNN layer (backend.py:93-119, transformer.py:170-176); keys, shapes and order as the
    reference's (the synthetic schema is pinned to it by tests/test_oracle_golden.py), strict loading, and the module
    still scripts and round-trips through torch.jit.save."""
    import io

    hypers = dict(default_hypers(), **VARIANTS[tag])
    be = PETBackend(hypers, [1, 6, 7, 8])
    be.add_output("energy", {"energy": [1]})
    sd = be.state_dict()
    schema = state_dict_schema(hypers, [1, 6, 7, 8], {"energy": 1})
    assert list(sd.keys()) == [k for k, _, _ in schema]
    for k, shape, _ in schema:
        assert tuple(sd[k].shape) == tuple(shape), k
    res = be.load_state_dict(opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert be.num_readout_layers == (hypers["num_gnn_layers"] if hypers["featurizer_type"] == "residual" else 1)
    mod = torch.jit.script(be)
    buf = io.BytesIO()
    torch.jit.save(mod, buf)
    buf.seek(0)
    back = torch.jit.load(buf)
    assert [tuple(a.shape) for a in back._params()] == [tuple(a.shape) for a in be._params()]
    assert len(be._params()) == len(list(be.parameters()))


def test_unsupported_variants_and_cpu_inputs_raise():
    with pytest.raises(ValueError, match="featurizer_type"):
        PETBackend(dict(default_hypers(), featurizer_type="convolutional"), [1, 6])
    with pytest.raises(ValueError, match="normalization"):
        PETBackend(dict(default_hypers(), normalization="BatchNorm"), [1, 6])
    with pytest.raises(ValueError, match="adaptive_cutoff_method"):
        PETBackend(dict(default_hypers(), num_neighbors_adaptive=16, adaptive_cutoff_method="bisection"), [1, 6])
    PETBackend(dict(default_hypers(), num_neighbors_adaptive=16), [1, 6])  # "solver"
    PETBackend(dict(default_hypers(), num_neighbors_adaptive=16, adaptive_cutoff_method="grid"), [1, 6])  # legacy method
    be = PETBackend(default_hypers(), [1, 6, 7, 8])
    be.add_output("energy", {"energy": [1]})
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU path"):
        be.preprocess(z(2, 3), z(0, dtype=torch.int32), z(0, dtype=torch.int32), torch.tensor([1, 6]),
                      z(1, 3, 3), z(0, 3, dtype=torch.int32), z(2, dtype=torch.long), 1.0)

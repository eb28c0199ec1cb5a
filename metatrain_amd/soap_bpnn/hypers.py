"""Default SOAP-BPNN model hypers (``soap_bpnn/documentation.py:53-120``)."""
import copy

_DEFAULTS = {
    "soap": {"max_angular": 6, "max_radial": 7, "cutoff": {"radius": 5.0, "width": 0.5}},
    "legacy": True,
    "bpnn": {"num_hidden_layers": 2, "num_neurons_per_layer": 32, "layernorm": True},
}


def default_hypers() -> dict:
    return copy.deepcopy(_DEFAULTS)

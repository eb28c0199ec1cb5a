"""Default PET hyper-parameters (mirror of ``ModelHypers``,
src/metatrain/pet/documentation.py:159-259) -- constants only."""

DEFAULT_MODEL_HYPERS = {
    "cutoff": 4.5,
    "num_neighbors_adaptive": None,
    "adaptive_cutoff_method": "solver",
    "cutoff_function": "Bump",
    "cutoff_width": 0.5,
    "cutoff_width_adaptive": 1.0,
    "d_pet": 128,
    "d_head": 128,
    "d_node": 256,
    "d_feedforward": 256,
    "num_heads": 8,
    "num_attention_layers": 2,
    "num_gnn_layers": 2,
    "normalization": "RMSNorm",
    "activation": "SwiGLU",
    "attention_temperature": 1.0,
    "transformer_type": "PreLN",
    "featurizer_type": "feedforward",
    "zbl": False,
    "long_range": {"enable": False},
    "system_conditioning": False,
    "max_charge": 10,
    "max_spin_multiplicity": 10,
}


def default_hypers() -> dict:
    out = dict(DEFAULT_MODEL_HYPERS)
    out["long_range"] = dict(DEFAULT_MODEL_HYPERS["long_range"])
    return out

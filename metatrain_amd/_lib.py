"""ctypes binding of libpet_hip.so (the C ABI declared in include/pet_hip.h).

There is no CPU fallback anywhere in this package: if the shared library is missing
or fails to load, importing/using the HIP path raises.
"""
import ctypes
import os

# torch MUST be imported before libpet_hip.so is loaded: device pointers and streams are
# exchanged with torch, so both have to run on the ONE HIP runtime torch ships
# (torch/lib/libamdhip64.so). Loading our library first would pull in /opt/rocm's copy and the
# process would end up with two runtimes ("no ROCm-capable device is detected").
import torch  # noqa: F401
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpet_hip.so")

PET_OK = 0
PET_CUTOFF_COSINE = 0
PET_CUTOFF_BUMP = 1

# every symbol include/pet_hip.h declares (tests check the library exports them all)
SYMBOLS = [
    "pet_last_error", "pet_version", "pet_hypers_supported",
    "pet_model_create", "pet_model_destroy", "pet_model_set_param", "pet_model_finalize",
    "pet_model_num_params",
    "pet_nl_workspace_bytes", "pet_nl_build", "pet_nl_batch_workspace_bytes", "pet_nl_build_batch",
    "pet_graph_workspace_bytes", "pet_graph_build", "pet_graph_destroy", "pet_graph_num_edges",
    "pet_graph_max_neighbors", "pet_graph_export_batch", "pet_graph_csr",
    "pet_graph_from_batch_workspace_bytes", "pet_graph_from_batch", "pet_model_block_properties",
    "pet_predict_scratch_floats", "pet_predict", "pet_predict_backward", "pet_geometry_backward",
    "pet_forward_workspace_bytes", "pet_forward_workspace_bytes_for", "pet_forward", "pet_aux_outputs", "pet_backward", "pet_backward_predict",
    "pet_backward_features", "pet_backward_geometry",
    "pet_model_num_readout_layers", "pet_forward_layers", "pet_backward_features_layers", "pet_graph_set_conditioning", "pet_graph_set_exchange",
    "pet_model_zero_grad", "pet_model_get_grad", "pet_train_workspace_bytes", "pet_train_workspace_bytes_for",
    "pet_train2_workspace_bytes_for", "pet_backward_train",
    "pet_model_get_param", "pet_model_flat_grad", "pet_adam_step", "pet_optimizer_state", "pet_model_tie_halves",
    "pet_train2_workspace_bytes", "pet_backward_train2", "pet_backward_train2_cell",
    "pet_sum_over_atoms",
    "pet_profile_enable", "pet_profile_select", "pet_profile_reset", "pet_profile_report", "pet_config_set",
]


SOAP_SYMBOLS = [
    "soap_model_create", "soap_model_destroy", "soap_model_feature_size", "soap_model_set_radial_table",
    "soap_model_set_param", "soap_model_finalize", "soap_workspace_bytes", "soap_forward", "soap_backward",
    "soap_model_zero_grad", "soap_train_workspace_bytes", "soap_train_gradients", "soap_model_get_grad",
    "soap_model_get_param", "soap_adam_step",
]
SOAP_MAX_L = 8


# pet_exchange_fn (include/pet_hip.h): the caller's collective of the per-layer exchange
EXCHANGE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_int)


class SoapHypers(ctypes.Structure):
    """Mirror of ``soap_hypers_t`` (include/soap_hip.h)."""

    _fields_ = [
        ("cutoff", c_float),
        ("cutoff_width", c_float),
        ("max_angular", c_int32),
        ("n_per_l", c_int32 * (SOAP_MAX_L + 1)),
        ("n_species", c_int32),
        ("n_channels", c_int32),
        ("legacy", c_int32),
        ("layernorm", c_int32),
        ("num_hidden_layers", c_int32),
        ("num_neurons_per_layer", c_int32),
    ]


class PetHypers(ctypes.Structure):
    """Mirror of ``pet_hypers_t``."""

    _fields_ = [
        ("cutoff", c_float),
        ("cutoff_width", c_float),
        ("cutoff_function", c_int32),
        ("d_pet", c_int32),
        ("d_head", c_int32),
        ("d_node", c_int32),
        ("d_feedforward", c_int32),
        ("num_heads", c_int32),
        ("num_attention_layers", c_int32),
        ("num_gnn_layers", c_int32),
        ("attention_temperature", c_float),
        ("nl_is_strict", c_int32),
        ("n_species", c_int32),
        ("max_atomic_number", c_int32),
        ("num_neighbors_adaptive", c_float),
        ("cutoff_width_adaptive", c_float),
        ("normalization", c_int32),
        ("transformer_type", c_int32),
        ("featurizer_type", c_int32),
        ("adaptive_cutoff_method", c_int32),
        ("system_conditioning", c_int32),
        ("max_charge", c_int32),
        ("max_spin_multiplicity", c_int32),
    ]


class PetHipError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load libpet_hip.so and declare the prototypes. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PetHipError(
            f"{LIB_PATH} not found: build it with `python -m metatrain_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    P = c_void_p
    lib.pet_last_error.restype = c_char_p
    lib.pet_version.restype = c_char_p
    lib.pet_hypers_supported.argtypes = [POINTER(PetHypers)]
    lib.pet_model_create.argtypes = [POINTER(PetHypers), POINTER(P)]
    lib.pet_model_destroy.argtypes = [P]
    lib.pet_model_destroy.restype = None
    lib.pet_model_set_param.argtypes = [P, c_char_p, P, c_int64, P]
    lib.pet_model_finalize.argtypes = [P, P]
    lib.pet_model_num_params.argtypes = [P]
    lib.pet_model_num_params.restype = c_int64
    lib.pet_nl_workspace_bytes.argtypes = [c_int64]
    lib.pet_nl_workspace_bytes.restype = c_int64
    lib.pet_nl_build.argtypes = [P, POINTER(c_float), POINTER(c_int32), c_int64, c_float, P, P, P,
                                 c_int64, POINTER(c_int64), P]
    lib.pet_nl_batch_workspace_bytes.argtypes = [c_int64, c_int64]
    lib.pet_nl_batch_workspace_bytes.restype = c_int64
    lib.pet_nl_build_batch.argtypes = [P, POINTER(c_float), POINTER(c_int32), POINTER(c_int64), c_int64, c_float, P, P, P,
                                       c_int64, POINTER(c_int64), P]
    lib.pet_graph_workspace_bytes.argtypes = [c_int64, c_int64]
    lib.pet_graph_workspace_bytes.restype = c_int64
    lib.pet_graph_build.argtypes = [P, P, P, P, P, P, P, P, c_int64, c_int64, c_int64, P, c_int64,
                                    POINTER(P), P]
    lib.pet_graph_destroy.argtypes = [P]
    lib.pet_graph_destroy.restype = None
    lib.pet_graph_num_edges.argtypes = [P]
    lib.pet_graph_num_edges.restype = c_int64
    lib.pet_graph_max_neighbors.argtypes = [P]
    lib.pet_graph_max_neighbors.restype = c_int32
    lib.pet_graph_export_batch.argtypes = [P] + [P] * 12 + [P]
    lib.pet_graph_csr.argtypes = [P, POINTER(P), POINTER(P), POINTER(P), POINTER(P)]
    lib.pet_graph_from_batch_workspace_bytes.argtypes = [c_int64, c_int64]
    lib.pet_graph_from_batch_workspace_bytes.restype = c_int64
    lib.pet_graph_from_batch.argtypes = [P] * 7 + [c_int64, c_int64, P, c_int64, POINTER(P), P]
    lib.pet_model_block_properties.argtypes = [P, c_char_p, c_int32, c_char_p]
    lib.pet_model_block_properties.restype = c_int32
    lib.pet_predict_scratch_floats.argtypes = [c_int64, c_int64]
    lib.pet_predict_scratch_floats.restype = c_int64
    lib.pet_predict.argtypes = [P, P, c_char_p, c_int32, c_char_p, P, P, P, P, P, P, P, P]
    lib.pet_predict_backward.argtypes = [P, P, c_char_p, c_int32, c_char_p, P, P, P, P, P, P, P, P, P]
    lib.pet_geometry_backward.argtypes = [P, P, P, P, P, P, P, P]
    lib.pet_forward_workspace_bytes.argtypes = [P, c_int64, c_int64]
    lib.pet_forward_workspace_bytes.restype = c_int64
    lib.pet_forward_workspace_bytes_for.argtypes = [P, P]
    lib.pet_forward_workspace_bytes_for.restype = c_int64
    lib.pet_forward.argtypes = [P, P, P, c_int64, c_int, P, P, P, P]
    lib.pet_aux_outputs.argtypes = [P, P, P, P, P, P, P, P]
    lib.pet_backward.argtypes = [P, P, P, c_int64, P, P, P, P]
    lib.pet_backward_predict.argtypes = [P, P, P, c_int64, P, P, P, P, P]
    lib.pet_backward_features.argtypes = [P, P, P, c_int64, P, P, P, P, P]
    lib.pet_backward_geometry.argtypes = [P, P, P, c_int64, P, P, P, P, P]
    lib.pet_graph_set_conditioning.argtypes = [P, P, P, P, c_int64]
    lib.pet_graph_set_exchange.argtypes = [P, P, c_int64, P, c_int64, P, P, EXCHANGE_FN, P]
    lib.pet_model_num_readout_layers.argtypes = [P]
    lib.pet_model_num_readout_layers.restype = c_int32
    lib.pet_forward_layers.argtypes = [P, P, P, c_int64, c_int, P, P, c_int32, P]
    lib.pet_backward_features_layers.argtypes = [P, P, P, c_int64, P, P, c_int32, P, P, P]
    lib.pet_model_zero_grad.argtypes = [P, P]
    lib.pet_model_get_grad.argtypes = [P, c_char_p, P, c_int64, P]
    lib.pet_model_get_param.argtypes = [P, c_char_p, P, c_int64, P]
    lib.pet_model_flat_grad.argtypes = [P, P, c_int64, c_int, P]
    lib.pet_adam_step.argtypes = [P, c_float, c_float, c_float, c_float, c_float, c_float, c_int64, P, P]
    lib.pet_optimizer_state.argtypes = [P, P, P, c_int64, c_int, P]
    lib.pet_model_tie_halves.argtypes = [P, c_char_p]
    lib.pet_train2_workspace_bytes.argtypes = [P, c_int64, c_int64]
    lib.pet_train2_workspace_bytes.restype = c_int64
    lib.pet_backward_train2.argtypes = [P, P, P, c_int64, P, c_int64, P, P, P, P, P]
    lib.pet_backward_train2_cell.argtypes = [P, P, P, c_int64, P, c_int64, P, P, P, P, P, P]
    lib.pet_train_workspace_bytes.argtypes = [P, c_int64, c_int64]
    lib.pet_train_workspace_bytes.restype = c_int64
    for fn in (lib.pet_train_workspace_bytes_for, lib.pet_train2_workspace_bytes_for):
        fn.argtypes = [P, P]
        fn.restype = c_int64
    lib.pet_backward_train.argtypes = [P, P, P, c_int64, P, P, P, P]
    lib.pet_sum_over_atoms.argtypes = [P, P, P, P]
    lib.pet_profile_enable.argtypes = [c_int]
    lib.pet_profile_select.argtypes = [c_char_p]
    lib.pet_profile_report.argtypes = [c_int, P, POINTER(c_double), POINTER(c_int64), POINTER(c_double),
                                       POINTER(c_double), POINTER(c_int)]
    lib.pet_config_set.argtypes = [c_char_p, c_int]
    lib.soap_model_create.argtypes = [POINTER(SoapHypers), POINTER(P)]
    lib.soap_model_destroy.argtypes = [P]
    lib.soap_model_destroy.restype = None
    lib.soap_model_feature_size.argtypes = [P]
    lib.soap_model_feature_size.restype = c_int64
    lib.soap_model_set_radial_table.argtypes = [P, P, c_int32, P]
    lib.soap_model_set_param.argtypes = [P, c_char_p, P, c_int64, P]
    lib.soap_model_finalize.argtypes = [P, P]
    lib.soap_workspace_bytes.argtypes = [P, c_int64, c_int64]
    lib.soap_workspace_bytes.restype = c_int64
    lib.soap_forward.argtypes = [P, P, P, c_int64, P, P, P]
    lib.soap_backward.argtypes = [P, P, P, c_int64, P, P, P, P]
    lib.soap_model_zero_grad.argtypes = [P, P]
    lib.soap_train_workspace_bytes.argtypes = [P, c_int64, c_int64]
    lib.soap_train_workspace_bytes.restype = c_int64
    lib.soap_train_gradients.argtypes = [P, P, P, c_int64, P, c_int64, P, P, P, P]
    lib.soap_model_get_grad.argtypes = [P, c_char_p, P, c_int64, P]
    lib.soap_model_get_param.argtypes = [P, c_char_p, P, c_int64, P]
    lib.soap_adam_step.argtypes = [P, c_float, c_float, c_float, c_float, c_int64, P]
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != PET_OK:
        msg = load().pet_last_error().decode(errors="replace")
        raise PetHipError(f"libpet_hip error {code}: {msg}")

"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/pet_hip.h declares, and its host-side argument checking behaves (no GPU calls)."""
import ctypes
import os
import re

import pytest
import torch

from metatrain_amd import _lib
from metatrain_amd import runtime as rt
from oracle import pet as opet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from metatrain_amd import build

        build.build(verbose=False)
    return _lib.load()


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "pet_hip.h")).read()
    declared = set(re.findall(r"\b(pet_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in pet_hip.h but not exported"
    assert b"gfx950" in lib.pet_version()


def test_soap_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "soap_hip.h")).read()
    declared = set(re.findall(r"\b(soap_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SOAP_SYMBOLS), declared ^ set(_lib.SOAP_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in soap_hip.h but not exported"


def test_hypers_struct_and_supported(lib):
    h = rt.hypers_struct(dict(opet.DEFAULT_HYPERS), [1, 6, 7, 8])
    assert lib.pet_hypers_supported(ctypes.byref(h)) == 1
    # any size is served since round 3 (the size-generic path, csrc/gen.hip): the reference's own minimal hypers
    # (pet/tests/test_basic.py:22-32), d_node == d_pet, sizes that are no multiple of a tile
    for sizes in (dict(d_pet=16, d_node=32, d_head=16, d_feedforward=32, num_heads=2),
                  dict(d_pet=1, d_node=1, d_head=1, d_feedforward=1, num_heads=1),
                  dict(d_pet=48, d_node=48, d_head=20, d_feedforward=72, num_heads=3)):
        hy = dict(opet.DEFAULT_HYPERS, **sizes)
        assert lib.pet_hypers_supported(ctypes.byref(rt.hypers_struct(hy, [1, 6]))) == 1
        m = rt.HipModel(hy, [1, 6])
        del m
    # what stays unsupported fails loudly with a message: d_pet not a multiple of num_heads, head dimension > 128
    for sizes in (dict(d_pet=100, num_heads=8), dict(d_pet=512, num_heads=2)):
        hy = dict(opet.DEFAULT_HYPERS, **sizes)
        assert lib.pet_hypers_supported(ctypes.byref(rt.hypers_struct(hy, [1, 6]))) == 0
        with pytest.raises(_lib.PetHipError, match="unsupported sizes"):
            rt.HipModel(hy, [1, 6])


def test_model_create_destroy_without_gpu(lib):
    m = rt.HipModel(dict(opet.DEFAULT_HYPERS), [1, 6, 7, 8])
    assert m.num_params == 0
    del m


def test_variants_outside_the_build_fail_loudly():
    """The variants of the reference's hypers are mapped onto pet_hypers_t (all built since round 2); a value the
    reference does not know raises like the reference does, and the C side refuses an enum it does not know."""
    for key, val, field in (("normalization", "LayerNorm", "normalization"), ("transformer_type", "PostLN", "transformer_type"),
                            ("featurizer_type", "residual", "featurizer_type")):
        assert getattr(rt.hypers_struct(dict(opet.DEFAULT_HYPERS, **{key: val}), [1, 6]), field) == 1
        with pytest.raises(ValueError, match=key):
            rt.hypers_struct(dict(opet.DEFAULT_HYPERS, **{key: "something else"}), [1, 6])
    grid = rt.hypers_struct(dict(opet.DEFAULT_HYPERS, num_neighbors_adaptive=12, adaptive_cutoff_method="grid"), [1, 6])
    assert grid.adaptive_cutoff_method == 1
    with pytest.raises(ValueError, match="adaptive_cutoff_method"):
        rt.hypers_struct(dict(opet.DEFAULT_HYPERS, adaptive_cutoff_method="bisection"), [1, 6])
    rt.hypers_struct(dict(opet.DEFAULT_HYPERS, activation="SiLU"), [1, 6])  # built: SwiGLU kernels, tied halves
    with pytest.raises(ValueError, match="Unknown activation flag"):  # transformer.py:342-346
        rt.hypers_struct(dict(opet.DEFAULT_HYPERS, activation="GELU"), [1, 6])
    with pytest.raises(ValueError, match="Unknown cutoff function type"):
        rt.hypers_struct(dict(opet.DEFAULT_HYPERS, cutoff_function="Step"), [1, 6])
    import ctypes

    lib = _lib.load()
    for field in ("normalization", "transformer_type", "featurizer_type", "adaptive_cutoff_method"):
        h = rt.hypers_struct(dict(opet.DEFAULT_HYPERS), [1, 6])
        setattr(h, field, 7)
        handle = ctypes.c_void_p()
        assert lib.pet_model_create(ctypes.byref(h), ctypes.byref(handle)) == -2, field  # PET_ERR_UNSUPPORTED
    cond = rt.hypers_struct(dict(opet.DEFAULT_HYPERS, system_conditioning=True, max_charge=4), [1, 6])
    assert (cond.system_conditioning, cond.max_charge, cond.max_spin_multiplicity) == (1, 4, 10)


def test_cpu_tensors_are_rejected_not_silently_computed():
    """The product has no CPU path: CPU tensors raise instead of falling back."""
    m = rt.HipModel(dict(opet.DEFAULT_HYPERS), [1, 6, 7, 8])
    params = opet.synthetic_params(dict(opet.DEFAULT_HYPERS), [1, 6, 7, 8], {"energy": 1})
    with pytest.raises(_lib.PetHipError, match="no CPU path"):
        m.load(params, "energy")
    with pytest.raises(_lib.PetHipError, match="no CPU path"):
        rt.neighbor_list(torch.zeros(4, 3), torch.eye(3), [True] * 3, 4.5)


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure; nothing under metatrain_amd/ may reference it."""
    pkg = os.path.join(ROOT, "metatrain_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f


def test_config_switches_of_removed_kernel_generations_are_rejected(lib):
    """Round 2 dropped the fp32-MFMA / bf16x6 TRR generations and two attention forms, round 4 the kernels that the
    software-pipelined edge-MLP / combination kernels and the shared-weight QKV kernel had superseded (k_emlp_h, k_emlp_bwd_h,
    k_emlp_bwd_r, k_comb_h, k_comb_bwd_h, the LDS-tile k_comb pair, k_qkv_h, k_qkv_hl): their switches are unknown keys
    (PET_ERR_ARGUMENT), the documented ones (include/pet_hip.h) are accepted."""
    for key in (b"bf16x6", b"f16x3", b"trr_persist", b"so_bf16x6", b"emlp_pipe", b"emlp_bwd_pipe", b"comb_pipe",
                b"comb_bwd_pipe", b"emlp_recompute", b"line_stores", b"lds_w", b"attn_fwd4", b"emlp_s_min", b"tile_mask",
                b"attn_lds", b"tile_f16x3", b"soap_fused", b"no_such_switch"):
        assert lib.pet_config_set(key, 0) == -3, key
    for key, default in ((b"trr", 1), (b"attn_fused", 3), (b"trr_compress", 3), (b"soap_packed", 1),
                         (b"node_planes", 1), (b"so_trr", 1), (b"so_f16x3", 1), (b"wgrad_bf16", 1), (b"side_stream", 1),
                         (b"center_fused", 1), (b"dxf_fused", 1), (b"node_split", 1), (b"sorted_shortcut", 1), (b"train_bf16", 0), (b"soap_ps_mfma", 1), (b"emlp_s", 1)):
        assert lib.pet_config_set(key, default) == 0, key

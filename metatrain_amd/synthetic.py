"""Synthetic inputs and weights for benchmarks and tests (no compute path here).

* :func:`random_box`: the periodic box of SURVEY §8(d) (cubic, rho = 0.05 atoms/A^3, positions
  U[0, L)^3, species uniform over {1, 6, 7, 8}, both from ``torch.Generator().manual_seed(seed)``).
* :func:`state_dict_schema` / :func:`synthetic_params`: the reference ``PETBackend`` state-dict
  layout filled by a documented per-key seeded generator, so that the same
  weights can be produced on any machine without shipping an 11.6 MB checkpoint.
"""
import math
from typing import Dict, List, Tuple

import torch


def random_box(n_atoms: int, seed: int, density: float = 0.05, dtype=torch.float32):
    gen = torch.Generator().manual_seed(seed)
    box = (n_atoms / density) ** (1.0 / 3.0)
    pos = torch.rand((n_atoms, 3), generator=gen, dtype=torch.float32) * box
    z = torch.tensor([1, 6, 7, 8])[torch.randint(0, 4, (n_atoms,), generator=gen)]
    cell = torch.eye(3, dtype=torch.float32) * box
    return pos.to(dtype), z.to(torch.int32), cell.to(dtype)


def state_dict_schema(hypers: dict, atomic_types: List[int], targets: Dict[str, int]):
    """Ordered ``(key, shape, kind)`` list of the reference ``PETBackend`` state dict."""
    d, dn, dh, dff = hypers["d_pet"], hypers["d_node"], hypers["d_head"], hypers["d_feedforward"]
    ns = len(atomic_types)
    out: List[Tuple[str, Tuple[int, ...], str]] = [
        ("species_to_species_index", (max(atomic_types) + 1,), "index")
    ]

    def lin(key, o, i):
        out.append((key + ".weight", (o, i), "linear_w"))
        out.append((key + ".bias", (o,), "linear_b"))

    layer_norm = hypers.get("normalization", "RMSNorm") == "LayerNorm"
    residual = hypers.get("featurizer_type", "feedforward") == "residual"
    n_readout = hypers["num_gnn_layers"] if residual else 1  # backend.py:93-119

    def norm(key, n):  # torch.nn.RMSNorm: weight; torch.nn.LayerNorm: weight, bias
        out.append((key + ".weight", (n,), "norm_w"))
        if layer_norm:
            out.append((key + ".bias", (n,), "norm_b"))

    for g in range(hypers["num_gnn_layers"]):
        for a in range(hypers["num_attention_layers"]):
            lp = f"gnn_layers.{g}.trans.layers.{a}"
            lin(lp + ".attention.input_linear", 3 * d, d)
            lin(lp + ".attention.output_linear", d, d)
            norm(lp + ".norm_attention", d)
            norm(lp + ".norm_mlp", d)
            lin(lp + ".mlp.w_in", (2 if hypers.get("activation", "SwiGLU") == "SwiGLU" else 1) * dff, d)
            lin(lp + ".mlp.w_out", d, dff)
            if dn != d:  # transformer.py:189-201: d_node == d_pet holds Identity modules (no parameters) instead
                lin(lp + ".center_contraction", d, dn)
                lin(lp + ".center_expansion", dn, d)
                norm(lp + ".norm_center_features", dn)
                lin(lp + ".center_mlp.w_in", (4 if hypers.get("activation", "SwiGLU") == "SwiGLU" else 2) * dn, dn)
                lin(lp + ".center_mlp.w_out", dn, 2 * dn)
        lin(f"gnn_layers.{g}.edge_embedder", d, 4)
        lin(f"gnn_layers.{g}.compress.0", d, (2 if g == 0 else 3) * d)
        lin(f"gnn_layers.{g}.compress.2", d, d)
        if g > 0:
            out.append((f"gnn_layers.{g}.neighbor_embedder.weight", (ns, d), "embedding"))
    if not residual:
        for g in range(hypers["num_gnn_layers"]):
            out.append((f"combination_norms.{g}.weight", (2 * d,), "norm_w"))
            out.append((f"combination_norms.{g}.bias", (2 * d,), "norm_b"))
        for g in range(hypers["num_gnn_layers"]):
            lin(f"combination_mlps.{g}.0", 2 * d, 2 * d)
            lin(f"combination_mlps.{g}.2", d, 2 * d)
    for l in range(n_readout):
        out.append((f"node_embedders.{l}.weight", (ns, dn), "embedding"))
    out.append(("edge_embedder.weight", (ns, d), "embedding"))
    if hypers.get("system_conditioning", False):  # conditioning.py:38-52 (created after the embedders, backend.py:121-130)
        out.append(("system_conditioning.charge_embedding.weight", (2 * hypers["max_charge"] + 1, dn), "embedding"))
        out.append(("system_conditioning.spin_multiplicity_embedding.weight", (hypers["max_spin_multiplicity"], dn), "embedding"))
        lin("system_conditioning.project.0", dn, 2 * dn)
        lin("system_conditioning.project.2", dn, dn)
    # a target maps to its number of properties (one block named like the target) or to {block: properties};
    # heads and last layers exist once per readout layer (backend.py:171-217)
    for t in targets:
        for l in range(n_readout):
            lin(f"node_heads.{t}.{l}.0", dh, dn)
            lin(f"node_heads.{t}.{l}.2", dh, dh)
    for t in targets:
        for l in range(n_readout):
            lin(f"edge_heads.{t}.{l}.0", dh, d)
            lin(f"edge_heads.{t}.{l}.2", dh, dh)
    for t, nprop in targets.items():
        for l in range(n_readout):
            for b, n in (nprop.items() if isinstance(nprop, dict) else [(t, nprop)]):
                lin(f"node_last_layers.{t}.{l}.{b}", n, dh)
    for t, nprop in targets.items():
        for l in range(n_readout):
            for b, n in (nprop.items() if isinstance(nprop, dict) else [(t, nprop)]):
                lin(f"edge_last_layers.{t}.{l}.{b}", n, dh)
    return out


def synthetic_params(hypers: dict, atomic_types: List[int], targets: Dict[str, int], seed: int = 0,
                     dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Key number ``n`` of the schema is drawn in float64 from
    ``torch.Generator().manual_seed(seed * 100003 + n)``: linear weights / biases
    ~ U(-1, 1)/sqrt(fan_in) (a bias uses its weight's fan_in), embeddings ~ U(-1, 1) sqrt(3),
    norm weights ~ 1 + 0.1 U(-1, 1), norm biases ~ 0.1 U(-1, 1); then cast to ``dtype``."""
    params: Dict[str, torch.Tensor] = {}
    fan_in = 1
    for n, (key, shape, kind) in enumerate(state_dict_schema(hypers, atomic_types, targets)):
        if kind == "index":
            idx = torch.full(shape, -1, dtype=torch.long)
            for i, z in enumerate(atomic_types):
                idx[z] = i
            params[key] = idx
            continue
        gen = torch.Generator().manual_seed(seed * 100003 + n)
        u = torch.rand(shape, generator=gen, dtype=torch.float64) * 2.0 - 1.0
        if kind == "linear_w":
            fan_in = shape[1]
            t = u / math.sqrt(fan_in)
        elif kind == "linear_b":
            t = u / math.sqrt(fan_in)
        elif kind == "embedding":
            t = u * math.sqrt(3.0)
        elif kind == "norm_w":
            t = 1.0 + 0.1 * u
        else:
            t = 0.1 * u
        params[key] = t.to(dtype)
    return params

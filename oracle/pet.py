"""
PET numerical-core oracle (torch, CPU) -- test infrastructure, never imported by
the product.

A functional restatement of the reference PET hot path on an **unpadded CSR edge
layout** (edges stably sorted by centre), with torch autograd providing dE/dR
exactly like the reference does (``utils/output_gradient.py:34-40``). It follows

* geometry / cutoff / filter      ``pet/modules/structures.py:206-316``,
                                  ``pet/modules/utilities.py:4-39``
* edge tokens + compress MLP      ``pet/modules/transformer.py:463-521``
* PreLN transformer layer         ``pet/modules/transformer.py:203-234``
* attention with log-cutoff bias  ``pet/modules/transformer.py:86-152, 565-589``
* SwiGLU feed-forward             ``pet/modules/transformer.py:39-50``
* feed-forward featuriser (ji gather + combination MLP)
                                  ``pet/modules/backend.py:496-587``
* heads / last layers / edge sum  ``pet/modules/backend.py:651-777, 468-481``

It is NOT the reference's code: the reference pads every atom's neighbour list to
the batch maximum (NEF layout) and lets pad keys into the softmax with a bias of
``log(1e-15)``; this restatement has no pads (difference ~1e-15 relative, SURVEY
Appendix B.1). Parameters are addressed by the reference's state-dict keys
 so that the same weights drive the reference, the oracle and the
HIP path.

Pinned by: the reference's regression energies
``pet/tests/test_regression.py:66-74`` (``tests/test_oracle_golden.py``) and the
golden vectors written by ``tests/golden/make_golden.py``.
"""

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import nef as _nef


DEFAULT_HYPERS = {
    # pet/documentation.py:159-259
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


def _adaptive_n_and_dn(r, d, centers, n_nodes, width, inv_rc, target):
    """``adaptive_cutoff.py:46-107``: smoothed neighbour count + cubic baseline and d/dr, closed form."""
    scaled = (d - (r[centers] - width)) / width
    active = (scaled > 0.0) & (scaled < 1.0)
    smaller = scaled <= 0.0
    s = math.pi * scaled.clamp(1e-6, 1.0 - 1e-6)
    sin_s = torch.sin(s)
    t = torch.tanh(torch.cos(s) / sin_s)
    f = torch.where(active, 0.5 * (1.0 + t), smaller.to(d.dtype))
    df = (0.5 * math.pi / width) * (1.0 - t * t) / (sin_s * sin_s) * active.to(d.dtype)
    n = torch.zeros(n_nodes, dtype=d.dtype).index_add(0, centers, f)
    dn = torch.zeros(n_nodes, dtype=d.dtype).index_add(0, centers, df)
    x = r * inv_rc
    return n + target * x**3, dn + 3.0 * target * x**2 * inv_rc


def adaptive_cutoffs_solver(centers, d, target: float, n_nodes: int, max_cutoff: float, width: float):
    """``adaptive_cutoff.py:110-229`` (``get_adaptive_cutoffs_solver``): ten Newton-bisection steps on the
    detached distances, then one implicit-function-theorem step that carries the gradient."""
    dd = d.detach()
    inv_rc = 1.0 / max_cutoff
    r_lo = torch.zeros(n_nodes, dtype=d.dtype)
    r_hi = torch.full((n_nodes,), max_cutoff, dtype=d.dtype)
    r = 0.5 * r_hi
    for _ in range(10):
        n, dn = _adaptive_n_and_dn(r, dd, centers, n_nodes, width, inv_rc, target)
        f = n - target
        below = f <= 0
        r_lo = torch.where(below, r, r_lo)
        r_hi = torch.where(below, r_hi, r)
        r_newton = r - f / dn.clamp_min(1e-6)
        inside = (r_newton >= r_lo) & (r_newton <= r_hi)
        r = torch.where(inside, r_newton, 0.5 * (r_lo + r_hi))
    _, dn_root = _adaptive_n_and_dn(r, dd, centers, n_nodes, width, inv_rc, target)
    per_edge = cutoff_bump(d, r[centers], width)  # the reference's cutoff_func with a per-edge cutoff
    n_res = torch.zeros(n_nodes, dtype=d.dtype).index_add(0, centers, per_edge) + target * (r * inv_rc) ** 3 - target
    return (r - n_res / dn_root.clamp_min(1e-6)).clamp(max_cutoff / 16.0, max_cutoff)


def grid_effective_num_neighbors(d, probes, centers, n_nodes: int, width: float):
    """``adaptive_cutoff.py:297-327``: smoothed neighbour count of every atom at every probe cutoff, ``[N, K]``."""
    weights = cutoff_bump(d.unsqueeze(0), probes.unsqueeze(1), width)  # [K, E]
    return torch.zeros((len(probes), n_nodes), dtype=d.dtype).index_add(1, centers, weights).T


def grid_gaussian_weights(n_eff, target: float):
    """``adaptive_cutoff.py:331-395`` with ``width=None``: Gaussian weights of the probes around the target count
    (+ the cubic baseline), width from the slope of the count along the probe axis, rows normalised."""
    x = torch.linspace(0, 1, n_eff.shape[1], dtype=n_eff.dtype)
    diff = n_eff - target + (target * x**3).unsqueeze(0)
    (slope,) = torch.gradient(diff, dim=-1)
    width_t = slope.abs().clamp_min(1e-12)
    logw = -0.5 * (diff / width_t) ** 2
    w = torch.exp(logw - logw.max())
    return w / w.sum(dim=1, keepdim=True)


def adaptive_cutoffs_grid(centers, d, target: float, n_nodes: int, max_cutoff: float, width: float,
                          min_cutoff: float = 0.5):
    """``adaptive_cutoff.py:232-294`` (``get_adaptive_cutoffs_grid``, the legacy method): smoothed neighbour counts on
    a grid of probe cutoffs (spacing width / 4), Gaussian weights around the target count, cutoff = weighted mean of the
    probes. Plain autograd for the gradient."""
    probes = torch.arange(min_cutoff, max_cutoff, width / 4.0, dtype=d.dtype)
    w = grid_gaussian_weights(grid_effective_num_neighbors(d, probes, centers, n_nodes, width), target)
    return probes @ w.T


def cutoff_bump(d: torch.Tensor, cutoff, width: float) -> torch.Tensor:
    """``pet/modules/utilities.py:4-22``."""
    s = (d - (cutoff - width)) / width
    s = s.clamp(1e-6, 1.0 - 1e-6)
    return 0.5 * (1.0 + torch.tanh(1.0 / torch.tan(math.pi * s)))


def cutoff_cosine(d: torch.Tensor, cutoff, width: float) -> torch.Tensor:
    """``pet/modules/utilities.py:25-39``."""
    s = ((d - (cutoff - width)) / width).clamp(0.0, 1.0)
    return 0.5 * (1.0 + torch.cos(math.pi * s))


def _linear(x, p, key):
    return torch.nn.functional.linear(x, p[key + ".weight"], p[key + ".bias"])


def _rmsnorm(x, weight):
    # torch.nn.RMSNorm(d) with eps=None -> finfo(dtype).eps
    eps = torch.finfo(x.dtype).eps
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * weight


def _norm(x, p, key):
    """norm_attention / norm_mlp / norm_center_features: torch.nn.RMSNorm (weight) or torch.nn.LayerNorm (weight, bias,
    eps 1e-5) -- ``getattr(nn, norm)(d)`` at transformer.py:176-186."""
    if key + ".bias" in p:
        return torch.nn.functional.layer_norm(x, (x.shape[-1],), p[key + ".weight"], p[key + ".bias"], 1e-5)
    return _rmsnorm(x, p[key + ".weight"])


def _swiglu_ff(x, p, prefix):
    y = _linear(x, p, prefix + ".w_in")
    if p[prefix + ".w_in.weight"].shape[0] == 2 * p[prefix + ".w_out.weight"].shape[1]:
        # activation = "SwiGLU", transformer.py:39-44: value first, gate second, sigmoid gate
        v, g = y.chunk(2, dim=-1)
        return _linear(v * torch.sigmoid(g), p, prefix + ".w_out")
    # activation = "SiLU", transformer.py:45-49
    return _linear(torch.nn.functional.silu(y), p, prefix + ".w_out")


class EdgeGraph:
    """CSR view of a (filtered) edge list, stably sorted by centre."""

    def __init__(self, centers, neighbors, cell_shifts, n_nodes):
        c = np.asarray(centers, dtype=np.int64)
        self.order = np.argsort(c, kind="stable")
        self.centers = c[self.order]
        self.neighbors = np.asarray(neighbors, dtype=np.int64)[self.order]
        self.shifts = np.asarray(cell_shifts, dtype=np.int64).reshape(-1, 3)[self.order]
        self.n_nodes = n_nodes
        self.counts = np.bincount(self.centers, minlength=n_nodes).astype(np.int64)
        self.rowptr = np.concatenate([[0], np.cumsum(self.counts)]).astype(np.int64)
        if len(c):
            self.rev = _nef.corresponding_edges(self.centers, self.neighbors, self.shifts)
        else:
            self.rev = np.zeros((0,), dtype=np.int64)


def edge_geometry(positions, cells, centers, neighbors, cell_shifts, system_indices):
    """``structures.py:206-221``: edge vectors and the ``|v| + 1e-15`` distance."""
    shifts = cell_shifts.to(positions.dtype)
    contrib = torch.einsum("ab,abc->ac", shifts, cells[system_indices[centers]])
    v = positions[neighbors] - positions[centers] + contrib
    d = torch.linalg.norm(v, dim=-1) + 1e-15
    return v, d


def _attention_bucketed(q, k, v, bias_center_first, graph: EdgeGraph, n_heads, scale):
    """Per-atom attention over tokens [centre; own edges], no padding.

    ``q, k, v``: ``(q_node [N,d], q_edge [E,d])`` pairs. ``bias_center_first``:
    per-edge key bias [E] (the centre key has bias log(1) = 0).
    Atoms are bucketed by neighbour count so that every bucket is a dense batch.
    """
    (qn, qe), (kn, ke), (vn, ve) = q, k, v
    d = qn.shape[-1]
    hd = d // n_heads
    out_n = torch.zeros_like(qn)
    out_e = torch.zeros_like(qe)
    counts = graph.counts
    for n in np.unique(counts):
        atoms = np.nonzero(counts == n)[0]
        a_t = torch.as_tensor(atoms)
        b = len(atoms)
        if n > 0:
            eidx = torch.as_tensor(graph.rowptr[atoms][:, None] + np.arange(n)[None, :])
        else:
            eidx = torch.zeros((b, 0), dtype=torch.long)

        def tok(xn, xe):
            t = torch.cat([xn[a_t][:, None, :], xe[eidx.reshape(-1)].reshape(b, n, d)], 1)
            return t.reshape(b, n + 1, n_heads, hd).permute(0, 2, 1, 3)

        qq, kk, vv = tok(qn, qe), tok(kn, ke), tok(vn, ve)
        bias = torch.cat(
            [
                torch.zeros(b, 1, dtype=qn.dtype),
                bias_center_first[eidx.reshape(-1)].reshape(b, n),
            ],
            1,
        )
        s = torch.matmul(qq, kk.transpose(-2, -1)) * scale + bias[:, None, None, :]
        o = torch.matmul(torch.softmax(s, dim=-1), vv)  # [b, h, T, hd]
        o = o.permute(0, 2, 1, 3).reshape(b, n + 1, d)
        out_n = out_n.index_add(0, a_t, o[:, 0])
        if n > 0:
            out_e = out_e.index_add(0, eidx.reshape(-1), o[:, 1:].reshape(b * n, d))
    return out_n, out_e


def pet_atomic_energies(
    params: Dict[str, torch.Tensor],
    hypers: dict,
    positions: torch.Tensor,
    cells: torch.Tensor,
    centers: torch.Tensor,
    neighbors: torch.Tensor,
    cell_shifts: torch.Tensor,
    species: torch.Tensor,
    system_indices: torch.Tensor,
    target: str = "energy",
    block: Optional[str] = None,
    return_features: bool = False,
    return_aux: bool = False,
    charge: Optional[torch.Tensor] = None,
    spin_multiplicity: Optional[torch.Tensor] = None,
):
    """Per-atom predictions ``[N, P]`` of the PET backend for one target.

    ``return_aux``: also the auxiliary per-atom outputs of ``pet/model.py:730-875`` -- ``feature``
    (``_get_output_features``, :750-755: node features | sum over edges of cutoff factor x edge features) and the
    target's last-layer features (``_get_output_last_layer_features``, :795-812: node-head hidden | cutoff-weighted
    sum of the edge-head hidden).

    Mirrors ``PETBackend.preprocess -> calculate_features -> predict``
    (``pet/modules/backend.py:238,344,420``) for the default variants
    (RMSNorm, SwiGLU, PreLN, feedforward featuriser, Bump/Cosine cutoff,
    no adaptive cutoff, non-strict neighbour list).
    """
    assert hypers["normalization"] in ("RMSNorm", "LayerNorm")
    assert hypers["activation"] in ("SwiGLU", "SiLU")
    assert hypers["transformer_type"] in ("PreLN", "PostLN")
    assert hypers["featurizer_type"] in ("feedforward", "residual")
    post_ln = hypers["transformer_type"] == "PostLN"
    residual = hypers["featurizer_type"] == "residual"
    assert hypers["num_neighbors_adaptive"] is None or hypers["adaptive_cutoff_method"] in ("solver", "grid")
    block = block or target
    cutoff, width = float(hypers["cutoff"]), float(hypers["cutoff_width"])
    n_heads = hypers["num_heads"]
    d_pet = hypers["d_pet"]
    n_nodes = positions.shape[0]
    p = params

    v_all, d_all = edge_geometry(
        positions, cells, centers.long(), neighbors.long(), cell_shifts, system_indices
    )
    pair_cut_all = None
    if hypers["num_neighbors_adaptive"] is not None:
        # structures.py:225-263: per-atom adaptive cutoffs, symmetrised per pair, then the mask
        solve = adaptive_cutoffs_grid if hypers["adaptive_cutoff_method"] == "grid" else adaptive_cutoffs_solver
        r_atom = solve(centers.long(), d_all, float(hypers["num_neighbors_adaptive"]), n_nodes, cutoff,
                       float(hypers["cutoff_width_adaptive"]))
        pair_cut_all = 0.5 * (r_atom[centers.long()] + r_atom[neighbors.long()])
        keep = torch.nonzero(d_all.detach() <= pair_cut_all.detach()).squeeze(-1)
    elif not bool(hypers["long_range"]["enable"]):
        # non-strict NL filter, structures.py:265-272
        keep = torch.nonzero(d_all.detach() <= cutoff).squeeze(-1)
    else:
        keep = torch.arange(len(d_all))
    graph = EdgeGraph(
        centers[keep].numpy(), neighbors[keep].numpy(), cell_shifts[keep].numpy(), n_nodes
    )
    sel = keep[torch.as_tensor(graph.order)]
    v = v_all[sel]
    d0 = d_all[sel]
    pair_cut = cutoff if pair_cut_all is None else pair_cut_all[sel]
    if hypers["cutoff_function"].lower() == "bump":
        fc = cutoff_bump(d0, pair_cut, width)
    else:
        fc = cutoff_cosine(d0, pair_cut, width)
    dist = torch.sqrt((v * v).sum(-1) + 1e-15)  # structures.py:330
    nbr = torch.as_tensor(graph.neighbors)
    ctr = torch.as_tensor(graph.centers)
    rev = torch.as_tensor(graph.rev)

    sp = p["species_to_species_index"][species.long()]
    sp_nbr = sp[nbr]

    h = p["node_embedders.0.weight"][sp]
    m = p["edge_embedder.weight"][sp_nbr]
    key_bias = torch.log(torch.clamp(fc, min=1e-15))  # transformer.py:109-110
    scale = 1.0 / (math.sqrt(d_pet // n_heads) * hypers["attention_temperature"])

    cond = None
    if hypers.get("system_conditioning", False):
        # conditioning.py:82-100: per-system embedding of charge and spin multiplicity, added to the node features
        # leaving every GNN layer (backend.py:517-545, :607-630); systems without the data: charge 0, multiplicity 1
        n_sys = cells.shape[0]
        q = torch.zeros(n_sys, dtype=torch.long) if charge is None else charge.long()
        sm = torch.ones(n_sys, dtype=torch.long) if spin_multiplicity is None else spin_multiplicity.long()
        x = torch.cat([p["system_conditioning.charge_embedding.weight"][q + hypers["max_charge"]],
                       p["system_conditioning.spin_multiplicity_embedding.weight"][sm - 1]], dim=-1)
        x = torch.nn.functional.silu(_linear(x, p, "system_conditioning.project.0"))
        cond = _linear(x, p, "system_conditioning.project.2")[system_indices.long()]

    node_feats, edge_feats = [], []
    for g in range(hypers["num_gnn_layers"]):
        pre = f"gnn_layers.{g}"
        if residual:  # backend.py:617: every GNN layer starts from its own node embedding
            h = p[f"node_embedders.{g}.weight"][sp]
        geo = torch.cat([v, dist[:, None]], dim=1)
        e = _linear(geo, p, pre + ".edge_embedder")
        if g == 0:
            tok = torch.cat([e, m], dim=1)
        else:
            tok = torch.cat([e, p[pre + ".neighbor_embedder.weight"][sp_nbr], m], dim=1)
        e = _linear(
            torch.nn.functional.silu(_linear(tok, p, pre + ".compress.0")),
            p,
            pre + ".compress.2",
        )
        for a in range(hypers["num_attention_layers"]):
            lp = f"{pre}.trans.layers.{a}"
            expanded = hypers["d_node"] != d_pet  # transformer.py:189-201
            c = _linear(h, p, lp + ".center_contraction") if expanded else h

            def attention(tn, te):
                qkv_n = _linear(tn, p, lp + ".attention.input_linear")
                qkv_e = _linear(te, p, lp + ".attention.input_linear")
                qn, kn, vn = qkv_n.split(d_pet, dim=-1)
                qe, ke, ve = qkv_e.split(d_pet, dim=-1)
                on, oe = _attention_bucketed((qn, qe), (kn, ke), (vn, ve), key_bias, graph, n_heads, scale)
                return (_linear(on, p, lp + ".attention.output_linear"), _linear(oe, p, lp + ".attention.output_linear"))

            if post_ln:  # transformer.py:236-262: norm AFTER each residual sum, the MLP on every token
                on, oe = attention(c, e)
                tn, te = _norm(c + on, p, lp + ".norm_attention"), _norm(e + oe, p, lp + ".norm_attention")
                tn = _norm(tn + _swiglu_ff(tn, p, lp + ".mlp"), p, lp + ".norm_mlp")
                e = _norm(te + _swiglu_ff(te, p, lp + ".mlp"), p, lp + ".norm_mlp")
                on = tn
            else:        # transformer.py:203-234
                on, oe = attention(_norm(c, p, lp + ".norm_attention"), _norm(e, p, lp + ".norm_attention"))
            if expanded:
                h = h + _linear(on, p, lp + ".center_expansion")
                h = h + _swiglu_ff(_norm(h, p, lp + ".norm_center_features"), p, lp + ".center_mlp")
            else:  # the node features leaving the layer ARE the centre token (transformer.py:221-227 skipped)
                h = on
            if not post_ln:
                e = e + oe
                e = e + _swiglu_ff(_norm(e, p, lp + ".norm_mlp"), p, lp + ".mlp")
        if cond is not None:
            h = h + cond
        if residual:     # backend.py:621-647: features of every layer are read out; messages are averaged with the
            node_feats.append(h)  # reversed ones, no combination MLP
            edge_feats.append(e)
            m = 0.5 * (m + e[rev])
            continue
        cat = torch.cat([e, e[rev]], dim=1)  # backend.py:559-570
        cat = torch.nn.functional.layer_norm(
            cat,
            (2 * d_pet,),
            p[f"combination_norms.{g}.weight"],
            p[f"combination_norms.{g}.bias"],
            1e-5,
        )
        upd = _linear(
            torch.nn.functional.silu(_linear(cat, p, f"combination_mlps.{g}.0")),
            p,
            f"combination_mlps.{g}.2",
        )
        m = m + e + upd
    if not residual:
        node_feats, edge_feats = [h], [m]

    silu = torch.nn.functional.silu
    atomic = None
    for l, (h, m) in enumerate(zip(node_feats, edge_feats)):  # backend.py:468-481: summed over readout layers
        nl = silu(_linear(silu(_linear(h, p, f"node_heads.{target}.{l}.0")), p, f"node_heads.{target}.{l}.2"))
        el = silu(_linear(silu(_linear(m, p, f"edge_heads.{target}.{l}.0")), p, f"edge_heads.{target}.{l}.2"))
        node_pred = _linear(nl, p, f"node_last_layers.{target}.{l}.{block}")
        edge_pred = _linear(el, p, f"edge_last_layers.{target}.{l}.{block}") * fc[:, None]
        contrib = node_pred.index_add(0, ctr, edge_pred)
        atomic = contrib if atomic is None else atomic + contrib
    if return_aux:
        def esum(x):
            return torch.zeros((n_nodes, x.shape[1]), dtype=x.dtype).index_add(0, ctr, x * fc[:, None])
        return atomic, torch.cat([h, esum(m)], dim=1), torch.cat([nl, esum(el)], dim=1)
    if return_features == "all":  # every readout layer (residual featuriser: one per GNN layer)
        return atomic, node_feats, edge_feats, graph
    if return_features:
        return atomic, h, m, graph
    return atomic


def energy_and_gradient(
    params, hypers, positions, cells, centers, neighbors, cell_shifts, species,
    system_indices, target="energy", create_graph=False, charge=None, spin_multiplicity=None,
):
    """Total energies per system and dE/dR (``utils/evaluate_model.py:128-133``)."""
    pos = positions.detach().clone().requires_grad_(True)
    atomic = pet_atomic_energies(
        params, hypers, pos, cells, centers, neighbors, cell_shifts, species,
        system_indices, target, charge=charge, spin_multiplicity=spin_multiplicity,
    )
    n_sys = cells.shape[0]
    energies = torch.zeros(n_sys, atomic.shape[1], dtype=atomic.dtype).index_add(
        0, system_indices.long(), atomic
    )
    (grad,) = torch.autograd.grad(energies.sum(), pos, create_graph=create_graph)
    if not create_graph:
        energies, atomic = energies.detach(), atomic.detach()
    return energies, grad, atomic


def batch_tensors(
    hypers, species_to_species_index, positions, cells, centers, neighbors, cell_shifts,
    species, system_indices,
) -> Dict[str, np.ndarray]:
    """numpy restatement of the 12-key ``batch_data`` dictionary
    (``pet/modules/backend.py:328-341`` <- ``structures.py:115-378``), for the
    fixed-cutoff path. Integer keys are bit-exact quantities; float keys are
    computed in the dtype of ``positions``."""
    pos = positions.detach()
    cutoff, width = float(hypers["cutoff"]), float(hypers["cutoff_width"])
    v, d0 = edge_geometry(
        pos, cells, centers.long(), neighbors.long(), cell_shifts, system_indices
    )
    n_nodes = pos.shape[0]
    pair_cut = None
    stats = np.full((n_nodes,), cutoff, dtype=pos.numpy().dtype)
    if hypers["num_neighbors_adaptive"] is not None:
        solve = adaptive_cutoffs_grid if hypers["adaptive_cutoff_method"] == "grid" else adaptive_cutoffs_solver
        r_atom = solve(centers.long(), d0, float(hypers["num_neighbors_adaptive"]), n_nodes, cutoff,
                       float(hypers["cutoff_width_adaptive"]))
        stats = r_atom.numpy()
        pair_cut = 0.5 * (r_atom[centers.long()] + r_atom[neighbors.long()])
        keep = torch.nonzero(d0 <= pair_cut).squeeze(-1)
        pair_cut = pair_cut[keep]
    elif not bool(hypers["long_range"]["enable"]):
        keep = torch.nonzero(d0 <= cutoff).squeeze(-1)
    else:
        keep = torch.arange(len(d0))
    centers_k, neighbors_k = centers[keep], neighbors[keep]
    shifts_k, v, d0 = cell_shifts[keep], v[keep], d0[keep]
    if hypers["cutoff_function"].lower() == "bump":
        fc = cutoff_bump(d0, cutoff if pair_cut is None else pair_cut, width)
    else:
        fc = cutoff_cosine(d0, cutoff if pair_cut is None else pair_cut, width)
    idx = _nef.reverse_neighbor_index(
        centers_k.numpy(), neighbors_k.numpy(), shifts_k.numpy(), n_nodes
    )
    nefi = torch.as_tensor(idx["nef_indices"])
    mask = torch.as_tensor(idx["padding_mask"])
    sp = species_to_species_index[species.long()]
    if len(keep):
        ev = v[nefi]
        sp_nbr = sp[neighbors_k.long()][nefi]
        fcn = torch.where(mask, fc[nefi], torch.zeros((), dtype=fc.dtype))
    else:
        m = nefi.shape[1]
        ev = torch.zeros((n_nodes, m, 3), dtype=pos.dtype)
        sp_nbr = torch.zeros((n_nodes, m), dtype=torch.long)
        fcn = torch.zeros((n_nodes, m), dtype=pos.dtype)
    ed = torch.sqrt((ev * ev).sum(-1) + 1e-15)
    return {
        "element_indices_nodes": sp.numpy(),
        "element_indices_neighbors": sp_nbr.numpy(),
        "edge_vectors": ev.numpy(),
        "edge_distances": ed.numpy(),
        "padding_mask": idx["padding_mask"],
        "reverse_neighbor_index": idx["reverse_neighbor_index"],
        "cutoff_factors": fcn.numpy(),
        "atomic_cutoffs_stats": stats,
        "centers": centers_k.numpy(),
        "neighbors": neighbors_k.numpy(),
        "nef_to_edges_neighbor": idx["nef_to_edges_neighbor"],
        "cell_shifts": shifts_k.numpy(),
    }


# --------------------------------------------------------------------------------------
# parameter construction
# --------------------------------------------------------------------------------------


def state_dict_schema(hypers: dict, atomic_types: List[int], targets: Dict[str, int]):
    """Ordered ``(key, shape, kind)`` list of the reference ``PETBackend`` state dict
    (creation order = SURVEY Appendix A RNG-order note).

    ``kind`` in {"index", "linear_w", "linear_b", "embedding", "norm_w", "norm_b"}.
    """
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
            lin(lp + ".mlp.w_in", (2 if hypers["activation"] == "SwiGLU" else 1) * dff, d)
            lin(lp + ".mlp.w_out", d, dff)
            if dn != d:  # transformer.py:189-201: d_node == d_pet holds Identity modules (no parameters) instead
                lin(lp + ".center_contraction", d, dn)
                lin(lp + ".center_expansion", dn, d)
                norm(lp + ".norm_center_features", dn)
                lin(lp + ".center_mlp.w_in", (4 if hypers["activation"] == "SwiGLU" else 2) * dn, dn)
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


def synthetic_params(
    hypers: dict, atomic_types: List[int], targets: Dict[str, int], seed: int = 0,
    dtype=torch.float32,
) -> Dict[str, torch.Tensor]:
    """Documented per-key seeded weight generator.

    Key number ``n`` (position in :func:`state_dict_schema`) is drawn in float64 from
    ``torch.Generator().manual_seed(seed * 100003 + n)``:
    linear weights / biases ~ U(-1, 1)/sqrt(fan_in) (biases use the weight's
    fan_in), embeddings ~ U(-1, 1) * sqrt(3), norm weights ~ 1 + 0.1 U(-1, 1),
    norm biases ~ 0.1 U(-1, 1); then cast to ``dtype``. It is independent of
    torch's module construction order, so fixtures stay valid across torch versions.
    """
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


def random_box(n_atoms: int, seed: int, density: float = 0.05, dtype=torch.float32):
    """Synthetic periodic box of SURVEY §8(d): cubic, rho = 0.05 / A^3, positions
    U[0, L)^3 then species uniform over {1, 6, 7, 8} from the same generator."""
    gen = torch.Generator().manual_seed(seed)
    box = (n_atoms / density) ** (1.0 / 3.0)
    pos = torch.rand((n_atoms, 3), generator=gen, dtype=torch.float32) * box
    z = torch.tensor([1, 6, 7, 8])[torch.randint(0, 4, (n_atoms,), generator=gen)]
    cell = torch.eye(3, dtype=torch.float32) * box
    return pos.to(dtype), z.to(torch.int32), cell.to(dtype)


def reference_init_params(
    hypers: dict, atomic_types: List[int], target: str, seed: int = 0
) -> Dict[str, torch.Tensor]:
    """Reproduce the weights the reference gets from ``torch.manual_seed(seed)``
    followed by ``PETBackend(hypers, types); add_output(target, {target: [1]})``
    by drawing torch's default initialisers in the same creation order
    (SURVEY Appendix A "RNG-order note": ``transformer.py:169-201,414-461``,
    ``backend.py:93-119,171-217``). Used to pin the oracle to the reference's own
    hard-coded regression energies (``pet/tests/test_regression.py:66-74``).
    """
    import random as _random

    _random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    d, dn, dh, dff = hypers["d_pet"], hypers["d_node"], hypers["d_head"], hypers["d_feedforward"]
    ns = len(atomic_types)
    p: Dict[str, torch.Tensor] = {}
    idx = torch.full((max(atomic_types) + 1,), -1, dtype=torch.long)
    for i, z in enumerate(atomic_types):
        idx[z] = i
    p["species_to_species_index"] = idx

    def lin(key, o, i):
        m = torch.nn.Linear(i, o)
        p[key + ".weight"] = m.weight.detach()
        p[key + ".bias"] = m.bias.detach()

    def emb(key, n, dim):
        p[key + ".weight"] = torch.nn.Embedding(n, dim).weight.detach()

    for g in range(hypers["num_gnn_layers"]):
        for a in range(hypers["num_attention_layers"]):
            lp = f"gnn_layers.{g}.trans.layers.{a}"
            lin(lp + ".attention.input_linear", 3 * d, d)
            lin(lp + ".attention.output_linear", d, d)
            p[lp + ".norm_attention.weight"] = torch.ones(d)
            p[lp + ".norm_mlp.weight"] = torch.ones(d)
            lin(lp + ".mlp.w_in", (2 if hypers["activation"] == "SwiGLU" else 1) * dff, d)
            lin(lp + ".mlp.w_out", d, dff)
            if dn != d:  # transformer.py:189-201
                lin(lp + ".center_contraction", d, dn)
                lin(lp + ".center_expansion", dn, d)
                p[lp + ".norm_center_features.weight"] = torch.ones(dn)
                lin(lp + ".center_mlp.w_in", (4 if hypers["activation"] == "SwiGLU" else 2) * dn, dn)
                lin(lp + ".center_mlp.w_out", dn, 2 * dn)
        lin(f"gnn_layers.{g}.edge_embedder", d, 4)
        lin(f"gnn_layers.{g}.compress.0", d, (2 if g == 0 else 3) * d)
        lin(f"gnn_layers.{g}.compress.2", d, d)
        if g > 0:
            emb(f"gnn_layers.{g}.neighbor_embedder", ns, d)
    for g in range(hypers["num_gnn_layers"]):
        p[f"combination_norms.{g}.weight"] = torch.ones(2 * d)
        p[f"combination_norms.{g}.bias"] = torch.zeros(2 * d)
    for g in range(hypers["num_gnn_layers"]):
        lin(f"combination_mlps.{g}.0", 2 * d, 2 * d)
        lin(f"combination_mlps.{g}.2", d, 2 * d)
    emb("node_embedders.0", ns, dn)
    emb("edge_embedder", ns, d)
    lin(f"node_heads.{target}.0.0", dh, dn)
    lin(f"node_heads.{target}.0.2", dh, dh)
    lin(f"edge_heads.{target}.0.0", dh, d)
    lin(f"edge_heads.{target}.0.2", dh, dh)
    lin(f"node_last_layers.{target}.0.{target}", 1, dh)
    lin(f"edge_last_layers.{target}.0.{target}", 1, dh)
    return p

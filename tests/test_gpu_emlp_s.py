"""GPU parity tests of the two-waves-per-SIMD edge-MLP kernels (csrc/pet_emlp_s.hip: k_emlp_s, and k_emlp_bwd_s, which
RECOMPUTES the SwiGLU pre-activations instead of reading saved ones; reference: pet/modules/transformer.py:39-50, 230-232)
at sizes the default policy hands to the pipelined kernels (``pet_config_set("emlp_s", 2)`` hands them every graph of at least two edge
rows; by default they serve graphs of at least 28 672 edges, i.e. the at-size tests). Through the C ABI, against goldens generated from
the reference and against the fp64 oracle. Bar: 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import nl as onl
from oracle import pet as opet

pytestmark = pytest.mark.gpu
TOL = 1e-5
TYPES = [1, 6, 7, 8]


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from metatrain_amd import runtime

    runtime.config_set("emlp_s", 2)
    yield runtime
    runtime.config_set("emlp_s", 1)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _model(rt, dev, hypers, params):
    m = rt.HipModel(hypers, TYPES)
    m.load({k: v.to(dev) for k, v in params.items()}, "energy")
    return m


def _graph(rt, model, dev, pos, cells, i, j, s, z, sysidx):
    return rt.HipGraph(model, pos.float().to(dev), cells.float().to(dev), torch.as_tensor(i).int().to(dev),
                       torch.as_tensor(j).int().to(dev), torch.as_tensor(s).int().to(dev), z.to(dev), sysidx.int().to(dev))


def _stages(rt, fw, seeds):
    rt.profile(True)
    try:
        atomic = fw.forward()
        grad = fw.backward(seeds if seeds is not None else torch.ones_like(atomic))
        torch.cuda.synchronize()
        return atomic, grad, {r["name"] for r in rt.profile_report()}
    finally:
        rt.profile(False)


@pytest.mark.parametrize("name", ["pet_default_box64.npz", "pet_default_box1000.npz", "pet_variant_layernorm_box64.npz"])
def test_forced_kernels_against_reference_goldens(rt, dev, golden_dir, name):
    """Per-atom energies and dE/dR of the reference (fp64): RMSNorm and LayerNorm models; edge counts that are not multiples
    of 32 or 128 (partial tiles, workgroups with idle waves); the last attention layer of a GNN layer takes the gather form."""
    g = dict(np.load(os.path.join(golden_dir, name)))
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    hypers = dict(opet.DEFAULT_HYPERS)
    if "layernorm" in name:
        hypers["normalization"] = "LayerNorm"
    params = {k[2:]: torch.tensor(v) for k, v in g.items() if k.startswith("p:")} or \
        opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    model = _model(rt, dev, hypers, params)
    graph = _graph(rt, model, dev, t("in_positions"), t("in_cells"), g["in_centers"], g["in_neighbors"], g["in_cell_shifts"],
                   t("in_species"), t("in_system_indices"))
    fw = rt.HipForward(model, graph)
    atomic, grad, stages = _stages(rt, fw, None)
    assert {"emlp", "emlp_bwd"} <= stages
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < TOL
    # against the pipelined kernels with saved pre-activations on the same graph: the two forms agree far inside the bar
    rt.config_set("emlp_s", 0)
    try:
        fw2 = rt.HipForward(model, graph)
        a2 = fw2.forward()
        g2 = fw2.backward(torch.ones_like(a2))
    finally:
        rt.config_set("emlp_s", 2)
    assert relmax(atomic.cpu().numpy(), a2.cpu().numpy()) < 2e-6
    assert relmax(grad.cpu().numpy(), g2.cpu().numpy()) < 5e-6
    # run-to-run bit identity (no atomics, fixed summation order)
    assert torch.equal(fw.forward(), atomic) and torch.equal(fw.backward(torch.ones_like(atomic)), grad)


@pytest.mark.parametrize("normalization,seed", [("RMSNorm", 41), ("LayerNorm", 42)])
def test_forced_kernels_mixed_systems_against_oracle(rt, dev, normalization, seed):
    """Three systems of different density in one batch, random norm weights (and biases), a random seed vector: per-atom
    energies and dE/dR against the fp64 oracle; seed linearity of the recomputing adjoint."""
    hypers = dict(opet.DEFAULT_HYPERS, normalization=normalization)
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    gen = torch.Generator().manual_seed(seed)
    for k in params:
        if ".norm_" in k:
            params[k] = params[k] + 0.3 * torch.randn(params[k].shape, generator=gen)
    model = _model(rt, dev, hypers, params)
    pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l, off = [], [], [], [], [], [], [], 0
    for k, (n, rho) in enumerate([(70, 0.003), (160, 0.05), (90, 0.085)]):
        box = (n / rho) ** (1.0 / 3.0)
        cell = torch.eye(3) * box
        pos = torch.rand(n, 3, generator=gen) * box
        z = torch.tensor(TYPES)[torch.randint(0, 4, (n,), generator=gen)]
        i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
        pos_l.append(pos); z_l.append(z.int()); cell_l.append(cell)
        i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off); s_l.append(torch.tensor(s).long())
        sys_l.append(torch.full((n,), k, dtype=torch.long))
        off += n
    pos, z, cells = torch.cat(pos_l), torch.cat(z_l), torch.stack(cell_l)
    i, j, s, sysidx = torch.cat(i_l), torch.cat(j_l), torch.cat(s_l), torch.cat(sys_l)
    graph = _graph(rt, model, dev, pos, cells, i, j, s, z, sysidx)
    fw = rt.HipForward(model, graph)
    w = (torch.rand(off, generator=gen) + 0.5)
    atomic, grad, stages = _stages(rt, fw, w.to(dev))
    assert {"emlp", "emlp_bwd"} <= stages
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    p = pos.double().clone().requires_grad_(True)
    a = opet.pet_atomic_energies(p64, hypers, p, cells.double(), i, j, s, z, sysidx)[:, 0]
    (g_ref,) = torch.autograd.grad((a * w.double()).sum(), p)
    assert relmax(atomic.cpu().numpy(), a.detach().numpy()) < TOL
    assert relmax(grad.cpu().numpy(), g_ref.numpy()) < TOL
    g2 = fw.backward((0.25 * w).to(dev)) + fw.backward((0.75 * w).to(dev))
    np.testing.assert_allclose(g2.cpu().numpy(), grad.cpu().numpy(), atol=2e-6 * float(grad.abs().max()))


@pytest.mark.parametrize("what,factor", [("mlp", 30.0), ("gains", 30.0), ("embeddings", 1e-4), ("heads", 30.0), ("heads", 0.02)])
def test_forced_kernels_operand_ranges(rt, dev, what, factor):
    """(Forced with the edge MLP's kernels: the edge head's k_head_s / k_head_bwd_s, csrc/pet_head_s.hip.)
    The kernels hold weights and normalised rows as fp16 planes of 64 x and the SwiGLU output / its adjoint as planes at
    scale 1: large MLP weights (hidden activations of magnitude 1e3), large norm gains and a tiny residual stream must neither
    overflow nor lose the low planes: finite results, within the bar or within 3 x what plain fp32 torch loses on the same
    weights."""
    hypers = dict(opet.DEFAULT_HYPERS)
    params = {k: v.clone() for k, v in opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32).items()}
    if what == "embeddings":
        for k in ("edge_embedder.weight", "node_embedders.0.weight", "gnn_layers.1.neighbor_embedder.weight"):
            params[k] *= factor
    elif what == "mlp":
        for k in params:
            if ".mlp.w_in." in k:
                params[k] *= factor
    elif what == "heads":  # k_head_s / k_head_bwd_s (pet_head_s.hip): un-normalised rows, every operand scaled per row
        for k in params:
            if k.startswith("edge_heads.") and k.endswith(".weight"):
                params[k] *= factor
    else:
        for k in params:
            if ".norm_mlp." in k:
                params[k] *= factor
    model = _model(rt, dev, hypers, params)
    pos, z, cell = opet.random_box(200, seed=7)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
    sysidx = torch.zeros(len(z), dtype=torch.long)
    graph = _graph(rt, model, dev, pos, cell[None], i, j, s, z, sysidx)
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad = fw.backward(torch.ones_like(atomic))
    assert torch.isfinite(atomic).all() and torch.isfinite(grad).all()

    def run(dtype):
        p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in params.items()}
        r = pos.to(dtype).clone().requires_grad_(True)
        a = opet.pet_atomic_energies(p, hypers, r, cell[None].to(dtype), torch.tensor(i), torch.tensor(j),
                                     torch.tensor(s).long(), z, sysidx)[:, 0]
        (gr,) = torch.autograd.grad(a.sum(), r)
        return a.detach().double().numpy(), gr.double().numpy()

    a64, g64 = run(torch.float64)
    a32, g32 = run(torch.float32)
    assert relmax(atomic.cpu().numpy(), a64) < max(TOL, 3 * relmax(a32, a64))
    assert relmax(grad.cpu().numpy(), g64) < max(TOL, 3 * relmax(g32, g64))


def test_adjoint_refuses_a_workspace_without_saved_preactivations_when_recomputation_is_off(rt, dev):
    """The forward notes on the graph, per workspace, that it did not write [v; g]; an adjoint that may not recompute them
    must refuse rather than read what nobody wrote."""
    hypers = dict(opet.DEFAULT_HYPERS)
    model = _model(rt, dev, hypers, opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32))
    pos, z, cell = opet.random_box(150, seed=3)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
    graph = _graph(rt, model, dev, pos, cell[None], i, j, s, z, torch.zeros(len(z), dtype=torch.long))
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    ref = fw.backward(torch.ones_like(atomic)).clone()
    fw.forward()
    rt.config_set("emlp_s", 0)
    try:
        with pytest.raises(rt.PetHipError):
            fw.backward(torch.ones_like(atomic))
    finally:
        rt.config_set("emlp_s", 2)
    fw.forward()
    assert torch.equal(fw.backward(torch.ones_like(atomic)), ref)


def test_large_graph_round5_kernels_against_the_kernels_they_replace(rt, dev):
    """A 20 000-atom box (380 k edges, more than 16 384 atoms): the default policy hands the edge MLP, the edge head, the compress
    adjoint (csrc/pet_emlp_s.hip, pet_head_s.hip, pet_compress_s.hip) and the three node-row Linear layers around the attention
    block (csrc/pet_center_s.hip: k_rowlin_s) to the two-workgroups-per-CU kernels; ``emlp_s = 0`` runs the kernels they replace.
    Both are within 1e-5 of the reference at the at-size tests' sizes; here they must agree with each other to 2e-6."""
    from metatrain_amd.synthetic import random_box

    hypers = dict(opet.DEFAULT_HYPERS)
    model = _model(rt, dev, hypers, opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32))
    n = 20000
    pos, z, cell = random_box(n, 5)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, hypers["cutoff"])
    graph = rt.HipGraph(model, pos.to(dev), cell.to(dev)[None], pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                        pairs[:, 2:5].contiguous(), z.to(dev), torch.zeros(n, dtype=torch.int32, device=dev))
    out = {}
    try:
        for mode in (0, 1):
            rt.config_set("emlp_s", mode)
            fw = rt.HipForward(model, graph)
            rt.profile(True)
            a = fw.forward().clone()
            g = fw.backward(torch.ones_like(a)).clone()
            torch.cuda.synchronize()
            rt.profile(False)
            out[mode] = (a, g)
            if mode == 1:  # run-to-run bit determinism of the large-graph kernels (no atomics, fixed reduction orders)
                a2 = fw.forward().clone()
                g2 = fw.backward(torch.ones_like(a2)).clone()
                assert torch.equal(a, a2) and torch.equal(g, g2)
    finally:
        rt.config_set("emlp_s", 2)  # (the module's fixture forces the kernels for the other tests)
    (a0, g0), (a1, g1) = out[0], out[1]
    assert torch.isfinite(a1).all() and torch.isfinite(g1).all()
    assert float((a0 - a1).abs().max() / a0.abs().max()) < 2e-6
    assert float((g0 - g1).abs().max() / g0.abs().max()) < 2e-6


def test_backward_after_a_forward_that_saved_nothing_is_refused(rt, dev, golden_dir):
    """``pet_forward(save_for_backward = 0)`` writes neither [v; g] nor the compress pre-activations; the workspace's
    forward record carries the save level and ``pet_backward`` on that workspace fails with PET_ERR_ARGUMENT instead of
    consuming unwritten buffers. A saving forward into the same workspace makes the adjoint valid again."""
    from metatrain_amd.runtime import PetHipError, _ptr, _stream, check

    g = dict(np.load(os.path.join(golden_dir, "pet_default_box64.npz")))
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    hypers = dict(opet.DEFAULT_HYPERS)
    model = _model(rt, dev, hypers, opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32))
    graph = _graph(rt, model, dev, t("in_positions"), t("in_cells"), g["in_centers"], g["in_neighbors"], g["in_cell_shifts"],
                   t("in_species"), t("in_system_indices"))
    fw = rt.HipForward(model, graph)
    atomic = torch.empty((graph.n_nodes,), dtype=torch.float32, device=dev)
    check(fw.lib.pet_forward(model.handle, graph.handle, _ptr(fw.workspace), fw.nbytes, 0, _ptr(atomic), None, None, _stream()))
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    with pytest.raises(PetHipError, match="save_for_backward = 0"):
        fw.backward(torch.ones_like(atomic))
    a2 = fw.forward()
    grad = fw.backward(torch.ones_like(a2))
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < TOL


def test_forced_kernels_on_a_deeper_model(rt, dev):
    """``k_emlp_bwd_s`` rebuilds the pass-through adjoint dY of the residual branch from its two fp16 planes (22 bits)
    instead of the fp32 rows, about 2.4e-7 of the row's largest entry per layer; the loss grows linearly with the depth. A model
    of 3 GNN x 3 attention layers (nine edge-MLP adjoints in a chain, against four of the default model) stays inside the same
    1e-5 bar against the fp64 oracle, with the fused attention adjoint forced as well."""
    hypers = dict(opet.DEFAULT_HYPERS, num_gnn_layers=3, num_attention_layers=3)
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    model = _model(rt, dev, hypers, params)
    pos, z, cell = opet.random_box(150, seed=77)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
    sysidx = torch.zeros(150, dtype=torch.long)
    rt.config_set("attn_fused", 7)
    try:
        graph = _graph(rt, model, dev, pos, cell[None], i, j, s, z, sysidx)
        fw = rt.HipForward(model, graph)
        atomic, grad, stages = _stages(rt, fw, None)
    finally:
        rt.config_set("attn_fused", 3)
    assert {"emlp", "emlp_bwd"} <= stages
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    _, g_ref, a_ref = opet.energy_and_gradient(p64, hypers, pos.double(), cell[None].double(), torch.tensor(i), torch.tensor(j),
                                               torch.tensor(s).long(), z, sysidx)
    assert relmax(atomic.cpu().numpy(), a_ref.numpy().ravel()) < TOL
    assert relmax(grad.cpu().numpy(), g_ref.numpy()) < TOL

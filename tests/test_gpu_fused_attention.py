"""GPU parity tests of the per-atom fused attention block (csrc/pet_ablk.hip; reference: pet/modules/transformer.py:86-152,
203-234) at sizes the default policy would hand to the three-kernel form (``pet_config_set("attn_fused", 7)`` forces the
fused kernels on any graph; by default they serve graphs of at least 3 840 attention tiles, i.e. the at-size tests).
Everything through the C ABI, against goldens generated from the reference and against the fp64 oracle. Bar: 1e-5."""
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

    runtime.config_set("attn_fused", 7)
    yield runtime
    runtime.config_set("attn_fused", 3)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _model(rt, dev, hypers, params):
    m = rt.HipModel(hypers, TYPES)
    m.load({k: v.to(dev) for k, v in params.items()}, "energy")
    return m


def _default_model(rt, dev):
    hypers = dict(opet.DEFAULT_HYPERS)
    return _model(rt, dev, hypers, opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32))


def _graph(rt, model, dev, pos, cells, i, j, s, z, sysidx):
    return rt.HipGraph(model, pos.float().to(dev), cells.float().to(dev), torch.as_tensor(i).int().to(dev),
                       torch.as_tensor(j).int().to(dev), torch.as_tensor(s).int().to(dev), z.to(dev), sysidx.int().to(dev))


@pytest.mark.parametrize("name", ["pet_default_box64.npz", "pet_default_box1000.npz"])
def test_fused_block_against_reference_goldens(rt, dev, golden_dir, name):
    """Per-atom energies and dE/dR of the reference (fp64) on the 64- and 1000-atom boxes; the box of 1000 atoms has paired
    tiles (two atoms of at most 32 tokens together), tiles of one atom and a few 64-slot tiles."""
    g = dict(np.load(os.path.join(golden_dir, name)))
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    model = _default_model(rt, dev)
    graph = _graph(rt, model, dev, t("in_positions"), t("in_cells"), g["in_centers"], g["in_neighbors"], g["in_cell_shifts"],
                   t("in_species"), t("in_system_indices"))
    fw = rt.HipForward(model, graph)
    rt.profile(True)  # the stage names say which kernels ran: the fused block, not the three-kernel form
    try:
        atomic = fw.forward()
        grad = fw.backward(torch.ones_like(atomic))
        torch.cuda.synchronize()
        stages = {r["name"] for r in rt.profile_report()}
    finally:
        rt.profile(False)
    assert {"attn_blk", "attn_blk_bwd"} <= stages and not stages & {"qkv", "attn_fwd", "oproj", "attn_bwd", "qkv_bwd"}, stages
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < TOL
    # a second build of the same graph pairs the same atoms: bit-identical results (the pairing decides the summation order)
    graph2 = _graph(rt, model, dev, t("in_positions"), t("in_cells"), g["in_centers"], g["in_neighbors"], g["in_cell_shifts"],
                    t("in_species"), t("in_system_indices"))
    fw2 = rt.HipForward(model, graph2)
    atomic2 = fw2.forward()
    assert torch.equal(atomic2, atomic) and torch.equal(fw2.backward(torch.ones_like(atomic2)), grad)


@pytest.mark.parametrize("normalization,seed", [("RMSNorm", 31), ("LayerNorm", 32)])
def test_fused_block_mixed_systems_against_oracle(rt, dev, normalization, seed):
    """Three systems in one batch -- isolated atoms and dimers (tiles of 1 - 3 tokens, many pairs), the reference density, a
    moderately dense one (some 64-slot tiles) -- for both norms of the transformer layer, with random norm weights (and biases)
    and a random seed vector: per-atom energies and dE/dR against the fp64 oracle, seed linearity of the adjoint."""
    hypers = dict(opet.DEFAULT_HYPERS, normalization=normalization)
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    gen = torch.Generator().manual_seed(seed)
    for k in params:
        if ".norm_" in k:
            params[k] = params[k] + 0.3 * torch.randn(params[k].shape, generator=gen)
    model = _model(rt, dev, hypers, params)
    pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l, off = [], [], [], [], [], [], [], 0
    for k, (n, rho) in enumerate([(70, 0.003), (160, 0.05), (90, 0.085)]):
        L = (n / rho) ** (1.0 / 3.0)
        cell = torch.eye(3) * L
        pos = torch.rand(n, 3, generator=gen) * L
        z = torch.tensor(TYPES)[torch.randint(0, 4, (n,), generator=gen)]
        i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
        pos_l.append(pos); z_l.append(z.int()); cell_l.append(cell)
        i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off); s_l.append(torch.tensor(s).long())
        sys_l.append(torch.full((n,), k, dtype=torch.long))
        off += n
    pos, z, cells = torch.cat(pos_l), torch.cat(z_l), torch.stack(cell_l)
    i, j, s, sysidx = torch.cat(i_l), torch.cat(j_l), torch.cat(s_l), torch.cat(sys_l)
    counts = np.bincount(i.numpy(), minlength=off)
    assert counts.min() == 0 and counts.max() <= 63 and (counts >= 32).any()
    graph = _graph(rt, model, dev, pos, cells, i, j, s, z, sysidx)
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    w = (torch.rand(off, generator=gen) + 0.5)
    grad = fw.backward(w.to(dev))
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    p = pos.double().clone().requires_grad_(True)
    a = opet.pet_atomic_energies(p64, hypers, p, cells.double(), i, j, s, z, sysidx)[:, 0]
    (g_ref,) = torch.autograd.grad((a * w.double()).sum(), p)
    assert relmax(atomic.cpu().numpy(), a.detach().numpy()) < TOL
    assert relmax(grad.cpu().numpy(), g_ref.numpy()) < TOL
    g2 = fw.backward((0.25 * w).to(dev)) + fw.backward((0.75 * w).to(dev))
    np.testing.assert_allclose(g2.cpu().numpy(), grad.cpu().numpy(), atol=2e-6 * float(grad.abs().max()))


@pytest.mark.parametrize("what,factor", [("stream", 3e3), ("gains", 30.0), ("embeddings", 1e-4)])
def test_fused_block_operand_ranges(rt, dev, what, factor):
    """The fused block holds every operand as two fp16 planes of 64 x (pet_ablk.hip): large output projections (the adjoint
    rescales dAO per atom), large norm gains (normalised rows of magnitude 30) and a tiny residual stream must neither
    overflow nor lose the low planes: no inf / nan, and energies / dE/dR within the bar or within 3 x what plain fp32 torch
    loses on the same weights."""
    hypers = dict(opet.DEFAULT_HYPERS)
    params = {k: v.clone() for k, v in opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32).items()}
    if what == "embeddings":
        for k in ("edge_embedder.weight", "node_embedders.0.weight", "gnn_layers.1.neighbor_embedder.weight"):
            params[k] *= factor
    elif what == "stream":
        for k in params:
            if k.endswith(("compress.2.weight", "compress.2.bias", "output_linear.weight", "output_linear.bias",
                           "w_out.weight", "w_out.bias", "center_expansion.weight", "center_expansion.bias")):
                params[k] *= factor
    else:
        for k in params:
            if ".norm_" in k or k.startswith("combination_norms"):
                params[k] *= factor
    model = _model(rt, dev, hypers, params)
    pos, z, cell = opet.random_box(300, seed=12)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, 4.5)
    i, j, s = torch.tensor(i), torch.tensor(j), torch.tensor(s).long()
    sysidx = torch.zeros(300, dtype=torch.long)
    graph = _graph(rt, model, dev, pos, cell[None], i, j, s, z, sysidx)
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad = fw.backward(torch.ones_like(atomic))
    assert torch.isfinite(atomic).all() and torch.isfinite(grad).all()

    def oracle(dtype):
        pd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in params.items()}
        p = pos.to(dtype).clone().requires_grad_(True)
        a = opet.pet_atomic_energies(pd, hypers, p, cell[None].to(dtype), i, j, s, z, sysidx)[:, 0]
        (g,) = torch.autograd.grad(a.sum(), p)
        return a.detach().double().numpy(), g.double().numpy()

    a64, g64 = oracle(torch.float64)
    a32, g32 = oracle(torch.float32)
    assert relmax(atomic.cpu().numpy(), a64) < max(TOL, 3 * relmax(a32, a64))
    assert relmax(grad.cpu().numpy(), g64) < max(2 * TOL, 3 * relmax(g32, g64))

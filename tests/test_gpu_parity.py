"""GPU parity tests (``-m gpu``): the HIP path, called through the C ABI, against
(1) golden vectors generated from the reference and (2) the CPU oracle on seeded inputs.

Bars (BASELINE.json north_star): neighbour / NEF indices bit-exact; energies and forces
within 1e-5 relative in fp32. "Relative" for forces is max|dF| / max|F| against the fp64
reference (the reference's own fp32 path sits at 2.8e-6 .. 4.2e-6).
"""
import os

import numpy as np
import pytest
import torch

from oracle import nl as onl
from oracle import pet as opet

pytestmark = pytest.mark.gpu

TOL = 1e-5
INT_KEYS = ["element_indices_nodes", "element_indices_neighbors", "padding_mask", "reverse_neighbor_index",
            "centers", "neighbors", "nef_to_edges_neighbor", "cell_shifts"]
FLOAT_KEYS = ["edge_vectors", "edge_distances", "cutoff_factors", "atomic_cutoffs_stats"]


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    from metatrain_amd import runtime

    return runtime


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(rt, dev):
    hypers = dict(opet.DEFAULT_HYPERS)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
    m = rt.HipModel(hypers, [1, 6, 7, 8])
    m.load({k: v.to(dev) for k, v in params.items()}, "energy")
    assert m.num_params == sum(v.numel() for k, v in params.items() if k != "species_to_species_index")
    return m


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _graph_from_golden(rt, model, g, dev):
    t = lambda k: torch.tensor(g[k]).to(dev)  # noqa: E731
    return rt.HipGraph(model, t("in_positions").float(), t("in_cells").float(), t("in_centers"),
                       t("in_neighbors"), t("in_cell_shifts"), t("in_species"), t("in_system_indices").int())


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("case", ["co2cell", "box64", "two_systems"])
def test_preprocess_matches_reference_batch_data(rt, model, dev, golden_dir, case):
    """PETBackend.preprocess: all 12 batch_data tensors; integers bit-exact."""
    g = _load(golden_dir, f"batch_{case}.npz")
    out = _graph_from_golden(rt, model, g, dev).export_batch()
    assert set(out) == set(INT_KEYS + FLOAT_KEYS)
    for k in INT_KEYS:
        got = out[k].cpu().numpy()
        assert got.dtype == g[k].dtype and got.shape == g[k].shape, k
        assert np.array_equal(got, g[k]), f"{k} is not bit-exact"
    for k in FLOAT_KEYS:
        np.testing.assert_allclose(out[k].cpu().numpy(), g[k], rtol=2e-6, atol=2e-6, err_msg=k)


def test_energy_features_and_gradient_box64(rt, model, dev, golden_dir):
    g = _load(golden_dir, "pet_default_box64.npz")
    graph = _graph_from_golden(rt, model, g, dev)
    fw = rt.HipForward(model, graph)
    atomic, nf, ef = fw.forward(want_features=True)
    grad = fw.backward(torch.ones_like(atomic))
    e = fw.sum_over_atoms(atomic)
    assert abs(float(e[0]) - g["energies_f64"][0, 0]) / abs(g["energies_f64"][0, 0]) < TOL
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    assert relmax(nf.cpu().numpy(), g["node_features_f64"]) < TOL
    rowptr = graph.csr()["rowptr"].cpu().numpy()
    efn, ref = ef.cpu().numpy(), g["edge_features_f64"]
    scale = np.abs(ref).max()
    for i in range(graph.n_nodes):  # CSR rows vs the real slots of the reference's padded grid
        n = rowptr[i + 1] - rowptr[i]
        assert np.abs(efn[rowptr[i]:rowptr[i + 1]] - ref[i, :n]).max() / scale < TOL
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < TOL
    # and against the reference's own fp32 numbers (same tolerance the reference uses for itself)
    np.testing.assert_allclose(atomic.cpu().numpy(), g["atomic_f32"].ravel(), rtol=1e-4, atol=1e-5)


def test_energy_and_forces_box1000(rt, model, dev, golden_dir):
    """BASELINE config 2: 1000-atom periodic box, fp32, energy + forces vs the reference."""
    g = _load(golden_dir, "pet_default_box1000.npz")
    graph = _graph_from_golden(rt, model, g, dev)
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad = fw.backward(torch.ones_like(atomic))
    e = float(atomic.double().sum())
    assert abs(e - g["energies_f64"][0, 0]) / abs(g["energies_f64"][0, 0]) < TOL
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < TOL
    rms = np.sqrt(((grad.cpu().numpy() - g["grad_f64"]) ** 2).mean() / (g["grad_f64"] ** 2).mean())
    assert rms < TOL


@pytest.mark.parametrize("n_atoms,seed,pbc", [(1000, 5, (True, True, True)), (300, 6, (True, True, False)),
                                               (40, 7, (False, False, False)), (17, 8, (True, True, True)),
                                               (200, 9, (True, False, False)), (200, 10, (False, False, True))])
def test_neighbor_list_set_equals_oracle(rt, dev, n_atoms, seed, pbc):
    pos, z, cell = opet.random_box(n_atoms, seed)
    if n_atoms == 17:  # cell thinner than the cutoff along one axis: periodic self-images
        cell = cell.clone()
        cell[2, 2] = 3.1
        cell[1, 0] = 1.3
        pos = pos - 4.0  # unwrapped inputs
    pairs, vec = rt.neighbor_list(pos.to(dev), cell, pbc, 4.5)
    i, j, s, d = onl.neighbor_list(pos.numpy(), cell.numpy(), pbc, 4.5)
    got = pairs.cpu().numpy()
    assert np.all(np.diff(got[:, 0]) >= 0), "pairs must be grouped by centre"
    order = np.lexsort((got[:, 4], got[:, 3], got[:, 2], got[:, 1], got[:, 0]))
    assert np.array_equal(got[order], np.column_stack([i, j, s])), "neighbour set differs"
    np.testing.assert_allclose(vec.cpu().numpy()[order], d, atol=2e-5)


def test_empty_and_isolated_systems(rt, model, dev):
    """Reference edge cases: isolated atoms (no edges) and a dissociated pair
    (pet/tests/test_functionality.py:79-159)."""
    pos = torch.tensor([[0.0, 0, 0], [30.0, 0, 0], [0, 30.0, 0]], device=dev)
    z = torch.tensor([1, 6, 8], dtype=torch.int32, device=dev)
    e0 = torch.zeros(0, dtype=torch.int32, device=dev)
    graph = rt.HipGraph(model, pos, torch.zeros(1, 3, 3, device=dev), e0, e0,
                        torch.zeros((0, 3), dtype=torch.int32, device=dev), z,
                        torch.zeros(3, dtype=torch.int32, device=dev))
    assert graph.n_edges == 0 and graph.max_neighbors == 0
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad = fw.backward(torch.ones_like(atomic))
    assert torch.isfinite(atomic).all() and float(grad.abs().max()) == 0.0
    hypers = model.hypers
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    ref = opet.pet_atomic_energies(params, hypers, pos.cpu().double(), torch.zeros(1, 3, 3, dtype=torch.float64),
                                   torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long),
                                   torch.zeros((0, 3), dtype=torch.long), z.cpu(), torch.zeros(3, dtype=torch.long))
    assert relmax(atomic.cpu().numpy(), ref.numpy().ravel()) < TOL


def test_zero_atoms_through_the_runtime(rt, model, dev):
    """No atom at all (pet/tests/test_functionality.py:79-103 at the C ABI): neighbour lists (single and batched, an empty
    system between two others), graph, forward, backward, per-system sums and the training reverse pass return zero-sized
    results instead of failing on null buffers; an empty system in the middle of a batch sums to zero."""
    hypers = model.hypers
    cell = torch.eye(3) * 10
    pairs, _ = rt.neighbor_list(torch.zeros((0, 3), device=dev), cell, [True] * 3, hypers["cutoff"])
    assert tuple(pairs.shape) == (0, 5)
    boxes = [opet.random_box(30, seed) for seed in (1, 2)]
    pos = torch.cat([boxes[0][0], boxes[1][0]]).to(dev)
    cells = torch.stack([boxes[0][2], torch.eye(3) * 5, boxes[1][2]])
    batched, _ = rt.neighbor_list_batch(pos, cells, [[True] * 3] * 3, [0, 30, 30, 60], hypers["cutoff"])
    single = sum(rt.neighbor_list(b[0].to(dev), b[2], [True] * 3, hypers["cutoff"])[0].shape[0] for b in boxes)
    assert batched.shape[0] == single
    e0 = torch.zeros(0, dtype=torch.int32, device=dev)
    graph = rt.HipGraph(model, torch.zeros((0, 3), device=dev), torch.zeros(1, 3, 3, device=dev), e0, e0,
                        torch.zeros((0, 3), dtype=torch.int32, device=dev), e0, e0)
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    assert atomic.shape == (0,) and fw.backward(torch.ones_like(atomic)).shape == (0, 3)
    assert fw.sum_over_atoms(atomic).tolist() == [0.0]
    ft = rt.HipForward(model, graph, train=True)
    model.zero_grad()
    ft.backward_train(torch.ones_like(ft.forward()))
    assert float(model.flat_grad().abs().max()) == 0.0
    # systems 0 and 2 hold the atoms, system 1 none
    sysidx = torch.cat([torch.zeros(30), torch.full((30,), 2)]).int().to(dev)
    z = torch.cat([boxes[0][1], boxes[1][1]]).to(dev)
    graph = rt.HipGraph(model, pos, cells.to(dev), batched[:, 0].contiguous(), batched[:, 1].contiguous(),
                        batched[:, 2:5].contiguous(), z, sysidx)
    fw = rt.HipForward(model, graph)
    energies = fw.sum_over_atoms(fw.forward())
    assert float(energies[1]) == 0.0 and float(energies[0]) != 0.0 and float(energies[2]) != 0.0


def test_batch_of_systems_vs_oracle_and_per_system_sum(rt, model, dev):
    """Several systems in one call (concatenate_structures layout), triclinic + cubic cells,
    a non-strict list (built with cutoff + 0.7) in shuffled edge order; checks energies per
    system, dE/dR and dE/dcell against the fp64 oracle."""
    hypers = model.hypers
    systems = []
    for k, (n, seed) in enumerate([(150, 21), (60, 22), (9, 23)]):
        pos, z, cell = opet.random_box(n, seed)
        if k == 1:
            cell = cell.clone(); cell[1, 0] = 1.7; cell[2, 1] = -0.9
        systems.append((pos, z, cell))
    pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l, off = [], [], [], [], [], [], [], 0
    for k, (pos, z, cell) in enumerate(systems):
        i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"] + 0.7)
        pos_l.append(pos); z_l.append(z); cell_l.append(cell)
        i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off); s_l.append(torch.tensor(s))
        sys_l.append(torch.full((len(z),), k, dtype=torch.int32)); off += len(z)
    pos, z, cells = torch.cat(pos_l), torch.cat(z_l), torch.stack(cell_l)
    i, j, s, sysidx = torch.cat(i_l), torch.cat(j_l), torch.cat(s_l), torch.cat(sys_l)
    perm = torch.randperm(len(i), generator=torch.Generator().manual_seed(1))
    i, j, s = i[perm], j[perm], s[perm]
    graph = rt.HipGraph(model, pos.to(dev), cells.to(dev), i.to(dev), j.to(dev), s.to(dev), z.to(dev), sysidx.to(dev))
    assert graph.n_edges < len(i)  # the d <= cutoff filter dropped the extra pairs
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad, gcell = fw.backward(torch.ones_like(atomic), want_cell_grad=True)
    e = fw.sum_over_atoms(atomic).cpu().numpy()

    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    p64 = pos.double().requires_grad_(True)
    c64 = cells.double().requires_grad_(True)
    ref_atomic = opet.pet_atomic_energies(params, hypers, p64, c64, i, j, s.long(), z, sysidx.long())
    ref_e = torch.zeros(3, dtype=torch.float64).index_add(0, sysidx.long(), ref_atomic[:, 0])
    gp, gc = torch.autograd.grad(ref_e.sum(), [p64, c64])
    assert relmax(e, ref_e.detach().numpy()) < TOL
    assert relmax(grad.cpu().numpy(), gp.numpy()) < TOL
    assert relmax(gcell.cpu().numpy(), gc.numpy()) < TOL


@pytest.mark.parametrize("spacing,above63", [(1.9, False), (1.62, True)])
def test_neighbour_count_buckets_against_oracle(rt, model, dev, spacing, above63):
    """One open-boundary system whose neighbour counts run from 4 to 57 (every bucketed attention
    instantiation -- 1..4 key tiles, the persistent LDS-DMA adjoint for <= 32 tokens and the per-atom staged one
    above -- serves some atoms of the same call) or to 80 (more than 63 neighbours: the general kernels)."""
    hypers = model.hypers
    gen = torch.Generator().manual_seed(11)
    grid = torch.stack(torch.meshgrid(*[torch.arange(5.0)] * 3, indexing="ij"), -1).reshape(-1, 3)
    dense = spacing * grid + 0.25 * (torch.rand(grid.shape, generator=gen) - 0.5)       # 125 atoms, 0.15 - 0.23 / A^3
    dilute = 10.0 + 2.9 * grid[:64 + 16] + 0.4 * (torch.rand((80, 3), generator=gen) - 0.5)
    pos = torch.cat([dense, dilute]).float()
    z = torch.tensor([1, 6, 7, 8])[torch.randint(0, 4, (len(pos),), generator=gen)].int()
    cell = torch.zeros(3, 3)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [False] * 3, hypers["cutoff"])
    i, j, s = torch.tensor(i), torch.tensor(j), torch.tensor(s)
    counts = torch.bincount(i, minlength=len(pos))
    assert (counts.max() > 63) == above63 and counts.min() < 15 and ((counts > 31) & (counts < 48)).any()
    assert above63 or (counts >= 48).any()
    sysidx = torch.zeros(len(pos), dtype=torch.int32)
    graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), i.to(dev), j.to(dev), s.to(dev), z.to(dev),
                        sysidx.to(dev))
    assert graph.max_neighbors == int(counts.max())
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad = fw.backward(torch.ones_like(atomic))
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    p64 = pos.double().requires_grad_(True)
    ref = opet.pet_atomic_energies(params, hypers, p64, cell[None].double(), i, j, s.long(), z, sysidx.long())
    (gp,) = torch.autograd.grad(ref.sum(), p64)
    assert relmax(atomic.cpu().numpy(), ref.detach().numpy().ravel()) < TOL
    assert relmax(grad.cpu().numpy(), gp.numpy()) < TOL


@pytest.mark.parametrize("n_atoms,seed", [(257, 5), (389, 6), (515, 7), (1031, 8)])
def test_ragged_tail_tiles_against_oracle(rt, model, dev, n_atoms, seed):
    """Edge and token counts that leave partially filled 32-row wave tiles and 128-row workgroups at the end of
    every row kernel (E mod 128 differs per case), with a non-uniform seed vector so that adjoint rows span several
    orders of magnitude (per-row power-of-two scaling of the f16x3 kernels): energies and dE/dR against the fp64
    oracle evaluated here. (Complements test_rotation_and_permutation_consistency, which is the test that caught a
    per-lane branch around spill code in the compress adjoint)"""
    hypers = model.hypers
    pos, z, cell = opet.random_box(n_atoms, seed)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
    i, j, s = torch.tensor(i), torch.tensor(j), torch.tensor(s)
    sysidx = torch.zeros(n_atoms, dtype=torch.int32)
    graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), i.to(dev), j.to(dev), s.to(dev), z.to(dev),
                        sysidx.to(dev))
    gen = torch.Generator().manual_seed(seed)
    w = (10.0 ** (4.0 * torch.rand(n_atoms, generator=gen) - 3.0)).float()  # 1e-3 .. 10
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad = fw.backward(w.to(dev))
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    p64 = pos.double().requires_grad_(True)
    ref = opet.pet_atomic_energies(params, hypers, p64, cell[None].double(), i, j, s.long(), z, sysidx.long())
    (gp,) = torch.autograd.grad((ref.ravel() * w.double()).sum(), p64)
    assert relmax(atomic.cpu().numpy(), ref.detach().numpy().ravel()) < TOL
    assert relmax(grad.cpu().numpy(), gp.numpy()) < TOL


def test_silu_activation_variant_against_reference_golden(rt, dev, golden_dir):
    """SURVEY §8(f)-4, activation = "SiLU" (transformer.py:32-49): the state dict has the reference's shapes (one
    projection, [d_ff, d]); energies and dE/dR against the reference's own fp64 output (make_golden.py --silu)."""
    from metatrain_amd.pet import default_hypers

    g = _load(golden_dir, "pet_silu_box64.npz")
    hypers = dict(default_hypers(), activation="SiLU")
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
    key = "gnn_layers.0.trans.layers.0.mlp.w_in.weight"
    assert params[key].shape == (hypers["d_feedforward"], hypers["d_pet"])
    m = rt.HipModel(hypers, [1, 6, 7, 8])
    m.load({k: v.to(dev) for k, v in params.items()}, "energy")
    graph = _graph_from_golden(rt, m, g, dev)
    fw = rt.HipForward(m, graph)
    atomic = fw.forward()
    grad = fw.backward(torch.ones_like(atomic))
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < TOL
    assert torch.equal(m.param(key).cpu(), params[key])  # the tie is invisible at the state-dict level


def test_feature_and_last_layer_feature_outputs(rt, model, dev):
    """the "feature" and "mtt::aux::energy_last_layer_features" outputs (pet/model.py:730-875),
    per atom, against the fp64 oracle."""
    hypers = model.hypers
    pos, z, cell = opet.random_box(203, 17)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
    i, j, s = torch.tensor(i), torch.tensor(j), torch.tensor(s)
    sysidx = torch.zeros(len(pos), dtype=torch.int32)
    graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), i.to(dev), j.to(dev), s.to(dev), z.to(dev),
                        sysidx.to(dev))
    fw = rt.HipForward(model, graph)
    atomic, nf, ef = fw.forward(want_features=True)
    feat, llf = fw.aux_outputs(nf, ef)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    ref, feat64, llf64 = opet.pet_atomic_energies(params, hypers, pos.double(), cell[None].double(), i, j, s.long(), z,
                                                  sysidx.long(), return_aux=True)
    assert feat.shape == (203, hypers["d_node"] + hypers["d_pet"]) and llf.shape == (203, 2 * hypers["d_head"])
    assert relmax(atomic.cpu().numpy(), ref.numpy().ravel()) < TOL
    assert relmax(feat.cpu().numpy(), feat64.numpy()) < TOL
    assert relmax(llf.cpu().numpy(), llf64.numpy()) < TOL
    only_feat, none = fw.aux_outputs(nf, ef, last_layer_features=False)
    assert none is None and torch.equal(only_feat, feat)


def test_device_collate_matches_cpu_batching(rt, model, dev):
    """neighbour lists + batching on the device (metatrain_amd.data.collate) against the CPU
    route (oracle NL per system, offsets added as concatenate_structures does): same pair set, same energies and
    gradient, targets concatenated in system order."""
    from metatrain_amd import data

    hypers = model.hypers
    systems = []
    for k, (n, seed, pbc) in enumerate([(90, 41, (True, True, True)), (40, 42, (False, False, False)),
                                        (70, 43, (True, True, True))]):
        pos, z, cell = opet.random_box(n, seed)
        if k == 2:
            cell = cell.clone(); cell[1, 0] = 1.3; cell[2, 1] = -0.7
        if not any(pbc):
            cell = torch.zeros(3, 3)
        systems.append((pos, z, cell, pbc))
    energies = [torch.tensor([float(k)]) for k in range(3)]
    forces = [torch.full((len(sy[1]), 3), float(k)) for k, sy in enumerate(systems)]
    batch = data.collate([(p.to(dev), z.to(dev), c.to(dev), pbc) for p, z, c, pbc in systems], hypers["cutoff"],
                         {"energy": energies, "forces": forces})
    assert batch["energy"].tolist() == [0.0, 1.0, 2.0] and batch["forces"].shape == (200, 3)
    ref_pairs, off = set(), 0
    i_l, j_l, s_l = [], [], []
    for pos, z, cell, pbc in systems:
        i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), list(pbc), hypers["cutoff"])
        i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off); s_l.append(torch.tensor(s))
        ref_pairs |= {(int(a) + off, int(b) + off, *map(int, sh)) for a, b, sh in zip(i, j, s)}
        off += len(z)
    got = torch.cat([batch["centers"][:, None], batch["neighbors"][:, None], batch["cell_shifts"]], 1).cpu().tolist()
    assert len(got) == len(ref_pairs) and {tuple(r) for r in got} == ref_pairs
    assert torch.equal(batch["system_indices"].cpu(), torch.cat([torch.full((len(sy[1]),), k, dtype=torch.int32)
                                                                 for k, sy in enumerate(systems)]))
    fw = rt.HipForward(model, data.graph_of(model, batch))
    atomic = fw.forward()
    grad = fw.backward(torch.ones_like(atomic))
    pos = torch.cat([sy[0] for sy in systems]); z = torch.cat([sy[1] for sy in systems])
    cells = torch.stack([sy[2] for sy in systems])
    graph_cpu = rt.HipGraph(model, pos.to(dev), cells.to(dev), torch.cat(i_l).to(dev), torch.cat(j_l).to(dev),
                            torch.cat(s_l).to(dev), z.to(dev), batch["system_indices"])
    fw2 = rt.HipForward(model, graph_cpu)
    a2 = fw2.forward()
    g2 = fw2.backward(torch.ones_like(a2))
    # the order of an atom's neighbours follows the input order, so sums differ in the last bits only
    assert relmax(atomic.cpu().numpy(), a2.cpu().numpy()) < 2e-6
    assert relmax(grad.cpu().numpy(), g2.cpu().numpy()) < 2e-6


def test_weighted_seed_vector_backward(rt, model, dev):
    """pet_backward with a non-trivial dL/d(atomic) seed (what autograd hands over when the
    loss is not the plain energy sum): linearity check against two unit-seed calls."""
    pos, z, cell = opet.random_box(120, 31)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5)
    graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), pairs[:, 0], pairs[:, 1], pairs[:, 2:5],
                        z.to(dev), torch.zeros(120, dtype=torch.int32, device=dev))
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    w1 = torch.rand(120, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    w2 = torch.ones(120, device=dev) - w1
    g1, g2 = fw.backward(w1), fw.backward(w2)
    g = fw.backward(torch.ones(120, device=dev))
    assert relmax((g1 + g2).cpu().numpy(), g.cpu().numpy()) < 5e-6
    # determinism: identical bits run to run (no float atomics anywhere)
    assert torch.equal(fw.backward(w1), g1)
    # seeds of any magnitude (a loss can hand over 1e-8 or 1e+6): the split-operand GEMMs scale adjoint rows by a
    # power of two, so the gradient is exactly homogeneous
    for scale in (2.0 ** -27, 2.0 ** 20):
        gs = fw.backward(w1 * scale)
        assert torch.equal(gs, g1 * scale), scale
    gs = fw.backward(w1 * 3.7e-9)
    assert relmax((gs / 3.7e-9).cpu().numpy(), g1.cpu().numpy()) < 2e-6


def test_rotation_and_permutation_consistency(rt, model, dev):
    """Size-independent properties at a larger size (2000 atoms): permuting the atoms permutes
    per-atom energies and gradients; translating by a lattice vector changes nothing."""
    n = 2000
    pos, z, cell = opet.random_box(n, 41)
    sysidx = torch.zeros(n, dtype=torch.int32, device=dev)

    def run(p, zz):
        pairs, _ = rt.neighbor_list(p.to(dev), cell, [True] * 3, 4.5)
        graph = rt.HipGraph(model, p.to(dev), cell[None].to(dev), pairs[:, 0], pairs[:, 1], pairs[:, 2:5],
                            zz.to(dev), sysidx)
        fw = rt.HipForward(model, graph)
        a = fw.forward()
        return a.cpu().numpy(), fw.backward(torch.ones_like(a)).cpu().numpy()

    a0, g0 = run(pos, z)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(2))
    a1, g1 = run(pos[perm], z[perm])
    assert relmax(a1, a0[perm.numpy()]) < 5e-6 and relmax(g1, g0[perm.numpy()]) < 5e-6
    a2, g2 = run(pos + cell[0] * 2 - cell[2], z)
    # the shifted fp32 positions themselves differ by ~1 ulp(100 A) = 8e-6 A: an input perturbation, not a kernel error
    assert relmax(a2, a0) < 1e-4 and relmax(g2, g0) < 1e-4
    # Newton's third law: the net force on a periodic box vanishes
    assert np.abs(g0.sum(0)).max() < 1e-3 * np.abs(g0).max()


def test_edge_order_and_the_sort_shortcut(rt, model, dev, golden_dir):
    """``pet_graph_build`` skips the radix sort of the edges when the list arrives ordered by centre with nothing to drop.
    The first build of a process asks the device (a 4-byte read-back); later builds assume what the build before them found
    and learn from their final read-back whether that was right -- a wrong "sorted" guess leaves the graph empty on the
    device and builds again with the sort. The calls below walk through every transition: sorted after sorted (guess
    right), shuffled after sorted (guess wrong, rebuilt), sorted after shuffled (sorted although it need not be), sorted
    again (guess right). The reference list of the 1 000-atom box is ordered: with the shortcut and with
    ``sorted_shortcut = 0`` the results are bit-identical (the stable sort is the identity); the same list shuffled, and
    with 40 edges beyond the cutoff appended (dropped by the non-strict filter), takes the sort and must meet the same bar
    against the reference golden."""
    g = _load(golden_dir, "pet_default_box1000.npz")
    assert (np.diff(g["in_centers"]) >= 0).all()
    t = lambda k: torch.tensor(g[k]).to(dev)  # noqa: E731

    def run(i, j, s):
        graph = rt.HipGraph(model, t("in_positions").float(), t("in_cells").float(), i, j, s, t("in_species"),
                            t("in_system_indices").int())
        fw = rt.HipForward(model, graph)
        a = fw.forward()
        return a, fw.backward(torch.ones_like(a)), graph

    a1, g1, graph1 = run(t("in_centers"), t("in_neighbors"), t("in_cell_shifts"))
    rt.config_set("sorted_shortcut", 0)
    try:
        a0, g0, graph0 = run(t("in_centers"), t("in_neighbors"), t("in_cell_shifts"))
    finally:
        rt.config_set("sorted_shortcut", 1)
    assert torch.equal(a0, a1) and torch.equal(g0, g1)
    for k in ("rowptr", "nbr", "rev"):
        assert torch.equal(graph0.csr()[k], graph1.csr()[k]), k
    gen = torch.Generator().manual_seed(3)
    n_e = len(g["in_centers"])
    # 40 extra pairs far beyond the cutoff (an atom with its own image 7 cells away, both directions): dropped
    far = torch.randint(0, 1000, (20,), generator=gen)
    i = torch.cat([torch.tensor(g["in_centers"]), far, far])
    j = torch.cat([torch.tensor(g["in_neighbors"]), far, far])
    sh = torch.zeros(20, 3, dtype=torch.tensor(g["in_cell_shifts"]).dtype)
    sh[:, 0] = 7
    s = torch.cat([torch.tensor(g["in_cell_shifts"]), sh, -sh])
    perm = torch.randperm(n_e + 40, generator=gen)
    a2, g2, graph2 = run(i[perm].to(dev), j[perm].to(dev), s[perm].to(dev))
    assert int(graph2.n_edges) == n_e
    assert relmax(a2.cpu().numpy(), g["atomic_f64"].ravel()) < TOL and relmax(g2.cpu().numpy(), g["grad_f64"]) < TOL
    assert relmax(a2.cpu().numpy(), a1.cpu().numpy()) < 2e-6
    for _ in range(2):  # the sorted list again: through the sort (the guess is now "unsorted"), then through the shortcut
        a3, g3, graph3 = run(t("in_centers"), t("in_neighbors"), t("in_cell_shifts"))
        assert torch.equal(a3, a1) and torch.equal(g3, g1)
        for k in ("rowptr", "nbr", "rev"):
            assert torch.equal(graph3.csr()[k], graph1.csr()[k]), k
    a4, g4, graph4 = run(i[perm].to(dev), j[perm].to(dev), s[perm].to(dev))  # and the wrong guess once more
    assert int(graph4.n_edges) == n_e and torch.equal(a4, a2) and torch.equal(g4, g2)


@pytest.mark.parametrize("drop", ["i<j", "i>j"])
def test_half_neighbour_list_is_reported(rt, model, dev, golden_dir, drop):
    """Every kept edge needs its (j, i, -S) partner (nef.py:88-166 has the same precondition). Only the edge of a pair with
    i <= j searches for it (k_reverse), so both ways of losing partners are covered: three edges with i < j removed leave
    their i > j partners unpaired, and the other way round; the count in the message is exact."""
    g = _load(golden_dir, "pet_default_box64.npz")
    i, j = np.asarray(g["in_centers"]), np.asarray(g["in_neighbors"])
    pick = np.nonzero(i < j if drop == "i<j" else i > j)[0][[0, 7, 19]]
    keep = np.ones(len(i), bool)
    keep[pick] = False
    t = lambda k: torch.tensor(g[k]).to(dev)  # noqa: E731
    with pytest.raises(Exception, match="3 kept edges have no reverse edge"):
        rt.HipGraph(model, t("in_positions").float(), t("in_cells").float(), t("in_centers")[keep], t("in_neighbors")[keep],
                    t("in_cell_shifts")[keep], t("in_species"), t("in_system_indices").int())


def _adaptive_model(rt, dev, method):
    hypers = dict(opet.DEFAULT_HYPERS, num_neighbors_adaptive=12, adaptive_cutoff_method=method,
                  cutoff_width_adaptive=1.0)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
    m = rt.HipModel(hypers, [1, 6, 7, 8])
    m.load({k: v.to(dev) for k, v in params.items()}, "energy")
    return m


@pytest.fixture(scope="module")
def adaptive_model(rt, dev):
    return _adaptive_model(rt, dev, "solver")


@pytest.mark.parametrize("case", ["box64", "two_systems"])
def test_adaptive_cutoff_grid_method_matches_reference(rt, dev, golden_dir, case):
    """The legacy "grid" method (adaptive_cutoff.py:232-395: probe-cutoff grid, Gaussian weights around the target count),
    kept by the reference so that existing checkpoints reload with their original behaviour: batch_data with integers
    bit-exact, per-atom cutoffs, E / per-atom E / dE/dR (the gradient runs through the weights of every probe) and
    dE/dcell against the oracle's autograd."""
    model = _adaptive_model(rt, dev, "grid")
    b = _load(golden_dir, f"batch_adaptive_grid_{case}.npz")
    graph = _graph_from_golden(rt, model, b, dev)
    out = graph.export_batch()
    for k in INT_KEYS:
        assert np.array_equal(out[k].cpu().numpy(), b[k]), f"{k} is not bit-exact"
    np.testing.assert_allclose(out["atomic_cutoffs_stats"].cpu().numpy(), b["atomic_cutoffs_stats"], rtol=5e-6)
    assert 3.0 < b["atomic_cutoffs_stats"].min() and b["atomic_cutoffs_stats"].max() < 4.2
    np.testing.assert_allclose(out["cutoff_factors"].cpu().numpy(), b["cutoff_factors"], rtol=3e-4, atol=3e-6)
    g = _load(golden_dir, f"pet_adaptive_grid_{case}.npz")
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad, gcell = fw.backward(torch.ones_like(atomic), want_cell_grad=True)
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    ref32 = relmax(g["grad_f32"], g["grad_f64"])
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < max(TOL, 3 * ref32)
    # dE/dcell from the oracle's autograd (fp64) on the same inputs
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    params = opet.synthetic_params(model.hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    cells = t("in_cells").clone().requires_grad_(True)
    a64 = opet.pet_atomic_energies(params, model.hypers, t("in_positions"), cells, t("in_centers"), t("in_neighbors"),
                                   t("in_cell_shifts").long(), t("in_species"), t("in_system_indices"))
    (gc64,) = torch.autograd.grad(a64.sum(), cells)
    assert relmax(gcell.cpu().numpy(), gc64.numpy()) < max(TOL, 3 * ref32)


@pytest.mark.parametrize("case", ["box64", "two_systems"])
def test_adaptive_cutoff_matches_reference(rt, adaptive_model, dev, golden_dir, case):
    """num_neighbors_adaptive = 12 (solver): batch_data with integers bit-exact, per-atom
    cutoffs, then E / per-atom E / dE/dR including the implicit-function gradient of the cutoffs."""
    b = _load(golden_dir, f"batch_adaptive_{case}.npz")
    graph = _graph_from_golden(rt, adaptive_model, b, dev)
    out = graph.export_batch()
    for k in INT_KEYS:
        got = out[k].cpu().numpy()
        assert got.shape == b[k].shape, k
        assert np.array_equal(got, b[k]), f"{k} is not bit-exact"
    np.testing.assert_allclose(out["atomic_cutoffs_stats"].cpu().numpy(), b["atomic_cutoffs_stats"], rtol=3e-6)
    assert b["atomic_cutoffs_stats"].max() < 4.2  # the adaptive cutoffs really are below the 4.5 A maximum
    np.testing.assert_allclose(out["cutoff_factors"].cpu().numpy(), b["cutoff_factors"], rtol=2e-4, atol=3e-6)
    for k in ("edge_vectors", "edge_distances"):
        np.testing.assert_allclose(out[k].cpu().numpy(), b[k], rtol=2e-6, atol=2e-6, err_msg=k)
    g = _load(golden_dir, f"pet_adaptive_{case}.npz")
    fw = rt.HipForward(adaptive_model, graph)
    atomic = fw.forward()
    grad = fw.backward(torch.ones_like(atomic))
    e = fw.sum_over_atoms(atomic).cpu().numpy()
    assert np.abs(e - g["energies_f64"].ravel()).max() / np.abs(g["energies_f64"]).max() < TOL
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < 1e-5  # the reference's own fp32 path: see below
    ref32 = relmax(g["grad_f32"], g["grad_f64"])
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < max(TOL, 3 * ref32)


@pytest.mark.parametrize("method", ["solver", "grid"])
def test_adaptive_cutoff_cell_gradient_and_second_order(rt, dev, golden_dir, method):
    """Adaptive cutoffs, both methods: dE/dcell (the implicit-function / probe-grid term reaches the cell through the shifts
    of ALL input edges) and the force-loss parameter gradients (tangent of the cutoffs in the second-order pass: so.hip
    k_adapt_rdot / k_adapt_rdot_grid) against autograd through the fp64 oracle."""
    adaptive_model = _adaptive_model(rt, dev, method)
    hypers = adaptive_model.hypers
    g = _load(golden_dir, "pet_adaptive_two_systems.npz")
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    graph = _graph_from_golden(rt, adaptive_model, g, dev)
    n = graph.n_nodes
    fw = rt.HipForward(adaptive_model, graph, train=True)
    atomic = fw.forward()
    ones = torch.ones(n, device=dev)
    grad, gcell = fw.backward(ones, want_cell_grad=True)
    gen = torch.Generator().manual_seed(2)
    u = torch.randn(n, 3, generator=gen)
    adaptive_model.zero_grad()
    tan = fw.backward_train2(ones, None, u.to(dev), want_tangent=True)
    got = adaptive_model.grads()

    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
    p64 = {k: (v if k == "species_to_species_index" else v.double().clone().requires_grad_(True))
           for k, v in params.items()}
    pos = t("in_positions").double().clone().requires_grad_(True)
    cells = t("in_cells").double().clone().requires_grad_(True)
    a_ref = opet.pet_atomic_energies(p64, hypers, pos, cells, t("in_centers"), t("in_neighbors"), t("in_cell_shifts"),
                                     t("in_species"), t("in_system_indices").long(), "energy")[:, 0]
    gp, gc = torch.autograd.grad(a_ref.sum(), [pos, cells], create_graph=True)
    assert relmax(grad.cpu().numpy(), gp.detach().numpy()) < 1e-5
    assert relmax(gcell.cpu().numpy(), gc.detach().numpy()) < 1e-5
    keys = [k for k in p64 if k != "species_to_species_index"]
    ref = dict(zip(keys, torch.autograd.grad((u.double() * gp).sum(), [p64[k] for k in keys], allow_unused=True)))
    lhs, rhs = float(tan.double().sum()), float((u.to(dev).double() * grad.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(rhs))
    worst = 0.0
    for k, r in ref.items():
        if r is None:
            continue
        scale = float(r.abs().max())
        if scale > 1e-12:
            worst = max(worst, float((got[k].cpu().double() - r).abs().max()) / scale)
    bar = 1e-5
    if method == "grid":   # Gaussian weights over 17 probe counts: the reference's own fp32 arithmetic as the yardstick
        p32 = {k: (v if k == "species_to_species_index" else v.float().clone().requires_grad_(True)) for k, v in params.items()}
        pos32 = t("in_positions").float().clone().requires_grad_(True)
        a32 = opet.pet_atomic_energies(p32, hypers, pos32, t("in_cells").float(), t("in_centers"), t("in_neighbors"),
                                       t("in_cell_shifts"), t("in_species"), t("in_system_indices").long(), "energy")[:, 0]
        (g32,) = torch.autograd.grad(a32.sum(), pos32, create_graph=True)
        r32 = dict(zip(keys, torch.autograd.grad((u * g32).sum(), [p32[k] for k in keys], allow_unused=True)))
        worst32 = max(float((r32[k].double() - r).abs().max()) / float(r.abs().max()) for k, r in ref.items()
                      if r is not None and float(r.abs().max()) > 1e-12)
        print("grid: HIP", worst, "torch fp32", worst32)
        bar = max(bar, 3 * worst32)
    assert worst < bar, worst


def test_unknown_species_is_an_error_not_an_out_of_bounds_read(rt, model, dev):
    pos, z, cell = opet.random_box(32, 3)
    z = z.clone()
    z[5] = 14  # silicon is not one of the model's atomic_types [1, 6, 7, 8]
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5)
    with pytest.raises(Exception, match="atomic_types"):
        rt.HipGraph(model, pos.to(dev), cell[None].to(dev), pairs[:, 0], pairs[:, 1], pairs[:, 2:5], z.to(dev),
                    torch.zeros(32, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("switch", ["trr", "attn_fused=0", "side_stream", "trr_compress", "node_planes",
                                    "node_planes=2", "center_fused", "dxf_fused", "node_split"])
def test_alternative_kernel_paths_agree(rt, model, dev, golden_dir, switch):
    """The fallbacks behind ``pet_config_set``: the LDS-tile kernels (trr=0, also the transformer-layer path of PostLN models;
    ), the three-kernel attention form (attn_fused=0: QKV / attention / projection with Q, K, V
    in HBM -- what the training forward and graphs with many atoms of more than 32 tokens run), a single stream
    (side_stream=0), the node-row kernels with 32 rows per workgroup (node_planes = 2; the default up to 16 384 atoms), the next layer's centre tokens
    by their own launch (center_fused = 0), dXF by its own k_dxf launch (dxf_fused = 0), one workgroup per node-row tile (node_split = 0) and the A/B switches of the round-2 kernels. Each must meet the same parity bar."""
    g = _load(golden_dir, "pet_default_box64.npz")
    graph = _graph_from_golden(rt, model, g, dev)
    key, _, val = switch.partition("=")
    default = {"trr_compress": 3, "attn_fused": 3}.get(key, 1)
    rt.config_set(key, int(val or 0))
    try:
        fw = rt.HipForward(model, graph)
        atomic = fw.forward()
        grad = fw.backward(torch.ones_like(atomic))
    finally:
        rt.config_set(key, default)
    assert relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()) < TOL
    assert relmax(grad.cpu().numpy(), g["grad_f64"]) < TOL


def test_fused_attention_block_on_a_dense_box(rt, model, dev):
    """The 64-slot instantiation of the fused attention block (pet_ablk.hip, atoms of 33 .. 64 tokens): by default a
    graph in which more than 5 % of the atoms need it runs the three-kernel form, so it is forced here (attn_fused = 7)
    on a box of 36 neighbours per atom; per-atom energies and dE/dR against the fp64 oracle and against the
    three-kernel form."""
    hypers = model.hypers
    pos, z, cell = opet.random_box(400, seed=5, density=0.095)
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
    counts = np.bincount(i, minlength=400)
    assert counts.max() <= 63 and (counts >= 32).mean() > 0.5  # most atoms in 64-slot tiles, none beyond
    sysidx = torch.zeros(400, dtype=torch.int32)
    out = {}
    try:
        for mode in (7, 0):
            rt.config_set("attn_fused", mode)  # before the graph: a small graph plans its attention tiles only when forced
            graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), torch.tensor(i).int().to(dev),
                                torch.tensor(j).int().to(dev), torch.tensor(s).int().to(dev), z.to(dev), sysidx.to(dev))
            fw = rt.HipForward(model, graph)
            atomic = fw.forward()
            out[mode] = (atomic.cpu().numpy(), fw.backward(torch.ones_like(atomic)).cpu().numpy())
    finally:
        rt.config_set("attn_fused", 3)
    assert relmax(out[7][0], out[0][0]) < 2e-6 and relmax(out[7][1], out[0][1]) < TOL
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)  # the fixture model's weights
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    _, g_ref, a_ref = opet.energy_and_gradient(p64, hypers, pos.double(), cell[None].double(), torch.tensor(i),
                                               torch.tensor(j), torch.tensor(s).long(), z, sysidx.long())
    assert relmax(out[7][0], a_ref.numpy().ravel()) < TOL and relmax(out[7][1], g_ref.numpy()) < TOL


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_mixed_density_batch_fills_every_attention_bucket(rt, model, dev, seed):
    """Three systems of very different density in ONE batch -- a dilute one with isolated atoms (1 token), the
    reference density (1-2 tiles of 16 tokens) and a dense one (3-4 tiles) -- so that the graph's four per-tile-count
    atom lists (graph.hip k_bucket_fill) are all non-empty: per-atom energies and dE/dR against the fp64 oracle, and
    bit-identical reruns (batches with more than 64 tokens per atom take the generic kernels: test above)."""
    hypers = model.hypers
    gen = torch.Generator().manual_seed(seed)
    pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l = [], [], [], [], [], [], []
    off = 0
    for k, (n, rho) in enumerate([(60, 0.004), (150, 0.05), (110, 0.125)]):
        L = (n / rho) ** (1.0 / 3.0)
        cell = torch.eye(3) * L
        pos = torch.rand(n, 3, generator=gen) * L
        z = torch.tensor([1, 6, 7, 8])[torch.randint(0, 4, (n,), generator=gen)]
        i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
        pos_l.append(pos); z_l.append(z); cell_l.append(cell)
        i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off); s_l.append(torch.tensor(s))
        sys_l.append(torch.full((n,), k, dtype=torch.int32))
        off += n
    pos, z, cells = torch.cat(pos_l), torch.cat(z_l), torch.stack(cell_l)
    i, j, s, sysidx = torch.cat(i_l), torch.cat(j_l), torch.cat(s_l), torch.cat(sys_l)
    deg = torch.bincount(i, minlength=off)
    assert int(deg.min()) == 0 and int(deg.max()) <= 63  # isolated atoms; at most four tiles, so the bucketed kernels run
    assert all(int(((deg + 1 + 15) // 16 == t).sum()) > 0 for t in (1, 2, 3, 4))
    graph = rt.HipGraph(model, pos.to(dev), cells.to(dev), i.to(dev), j.to(dev), s.to(dev), z.to(dev), sysidx.to(dev))
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    w = (0.5 + torch.rand(off, generator=gen)).float()
    grad = fw.backward(w.to(dev))
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    p64 = pos.double().requires_grad_(True)
    ref = opet.pet_atomic_energies(params, hypers, p64, cells.double(), i, j, s.long(), z, sysidx.long())
    (gp,) = torch.autograd.grad((ref.ravel() * w.double()).sum(), p64)
    assert relmax(atomic.cpu().numpy(), ref.detach().numpy().ravel()) < TOL
    assert relmax(grad.cpu().numpy(), gp.numpy()) < TOL
    a2 = fw.forward()
    g2 = fw.backward(w.to(dev))
    assert torch.equal(atomic, a2) and torch.equal(grad, g2)  # list order inside a bucket is arbitrary, results are not


@pytest.mark.parametrize("pbc", [(True, False, False), (False, False, True), (False, True, True)])
def test_partly_periodic_triclinic_cells_against_oracle(rt, model, dev, pbc):
    """Wires and slabs: one or two periodic directions of a triclinic cell, atoms partly outside the cell -- device
    neighbour list, energies, dE/dR and dE/dcell against the fp64 oracle (the round-2 randomized sweep ran hundreds of
    such cases once; this keeps three in the suite)."""
    hypers = model.hypers
    rng = np.random.default_rng(sum(pbc) * 7 + pbc.index(True))
    n = 150
    cell = np.array([[14.0, 1.5, -0.8], [0.9, 12.0, 1.1], [-1.2, 0.7, 16.0]])
    pos = torch.tensor((rng.random((n, 3)) * 1.2 - 0.1) @ cell, dtype=torch.float32)
    cells = torch.tensor(cell, dtype=torch.float32)[None]
    z = torch.tensor(rng.choice([1, 6, 7, 8], n))
    pairs, _ = rt.neighbor_list(pos.to(dev), cells[0], list(pbc), hypers["cutoff"])
    i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cells[0].double().numpy(), list(pbc), hypers["cutoff"])
    got = pairs.cpu().numpy()
    order = np.lexsort((got[:, 4], got[:, 3], got[:, 2], got[:, 1], got[:, 0]))
    assert np.array_equal(got[order], np.column_stack([i, j, s]))
    sysidx = torch.zeros(n, dtype=torch.int32)
    graph = rt.HipGraph(model, pos.to(dev), cells.to(dev), pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                        pairs[:, 2:5].contiguous(), z.to(dev), sysidx.to(dev))
    fw = rt.HipForward(model, graph)
    atomic = fw.forward()
    grad, gcell = fw.backward(torch.ones_like(atomic), want_cell_grad=True)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    p64 = pos.double().requires_grad_(True)
    c64 = cells.double().requires_grad_(True)
    ref = opet.pet_atomic_energies(params, hypers, p64, c64, torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), z,
                                   sysidx.long())
    gp, gc = torch.autograd.grad(ref.sum(), [p64, c64])
    assert relmax(atomic.cpu().numpy(), ref.detach().numpy().ravel()) < TOL
    assert relmax(grad.cpu().numpy(), gp.numpy()) < TOL
    assert relmax(gcell.cpu().numpy(), gc.numpy()) < TOL

"""CPU tests: the oracle (``oracle/``) against the golden vectors generated from the
reference (``tests/golden/make_golden.py``) and against the reference's own
hard-coded regression energies (``pet/tests/test_regression.py:66-74``)."""

import os

import numpy as np
import pytest
import torch

from oracle import nef as onef
from oracle import nl as onl
from oracle import pet as opet


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def test_oracle_reproduces_reference_regression_energies(golden_dir):
    """Seed-0 default-init PET on the first five QM9 frames: the five numbers
    hard-coded in the reference's test_regression.py."""
    g = _load(golden_dir, "qm9_first5.npz")
    hypers = dict(opet.DEFAULT_HYPERS)
    params = opet.reference_init_params(hypers, [1, 6, 7, 8], "mtt::U0", seed=0)
    got = []
    for k in range(5):
        z, xyz = g[f"z{k}"], g[f"pos{k}"]
        i, j, s, _ = onl.neighbor_list(xyz, np.zeros((3, 3)), [False] * 3, hypers["cutoff"])
        e, _, _ = opet.energy_and_gradient(
            params, hypers, torch.tensor(xyz, dtype=torch.float32), torch.zeros(1, 3, 3),
            torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), torch.tensor(z),
            torch.zeros(len(z), dtype=torch.long), target="mtt::U0",
        )
        got.append(float(e[0, 0]))
    # the reference's own tolerance: torch.testing.assert_close fp32 defaults
    np.testing.assert_allclose(got, g["expected_reference_test"], rtol=1.3e-6, atol=1e-5)
    np.testing.assert_allclose(got, g["reference_import_seed0"], rtol=1.3e-6, atol=2e-6)


INT_KEYS = [
    "element_indices_nodes", "element_indices_neighbors", "padding_mask",
    "reverse_neighbor_index", "centers", "neighbors", "nef_to_edges_neighbor", "cell_shifts",
]
FLOAT_KEYS = ["edge_vectors", "edge_distances", "cutoff_factors", "atomic_cutoffs_stats"]


@pytest.mark.parametrize("case", ["co2cell", "box64", "two_systems"])
def test_oracle_batch_tensors_match_reference(golden_dir, case):
    g = _load(golden_dir, f"batch_{case}.npz")
    hypers = dict(opet.DEFAULT_HYPERS)
    idx = torch.full((9,), -1, dtype=torch.long)
    for n, z in enumerate([1, 6, 7, 8]):
        idx[z] = n
    got = opet.batch_tensors(
        hypers, idx, torch.tensor(g["in_positions"]), torch.tensor(g["in_cells"]),
        torch.tensor(g["in_centers"]), torch.tensor(g["in_neighbors"]),
        torch.tensor(g["in_cell_shifts"]), torch.tensor(g["in_species"]),
        torch.tensor(g["in_system_indices"]),
    )
    for k in INT_KEYS:
        assert got[k].shape == g[k].shape, k
        assert np.array_equal(got[k], g[k]), f"integer key {k} not bit-exact"
    for k in FLOAT_KEYS:
        np.testing.assert_allclose(got[k], g[k], rtol=2e-6, atol=2e-6, err_msg=k)


def _run_oracle(g, hypers, params, dtype):
    return opet.energy_and_gradient(
        params, hypers, torch.tensor(g["in_positions"], dtype=dtype),
        torch.tensor(g["in_cells"], dtype=dtype), torch.tensor(g["in_centers"]),
        torch.tensor(g["in_neighbors"]), torch.tensor(g["in_cell_shifts"]).long(),
        torch.tensor(g["in_species"]), torch.tensor(g["in_system_indices"]),
    )


@pytest.mark.parametrize("dtype,sfx,tol", [(torch.float64, "f64", 1e-10), (torch.float32, "f32", 2e-5)])
def test_oracle_tiny_energy_and_gradient(golden_dir, dtype, sfx, tol):
    g = _load(golden_dir, "pet_tiny.npz")
    hypers = dict(opet.DEFAULT_HYPERS, d_pet=16, d_node=32, d_head=16, d_feedforward=32, num_heads=2)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, dtype)
    for k, v in params.items():  # the committed state dict IS the generator's output
        np.testing.assert_allclose(v.double().numpy(), g["param::" + k], rtol=1e-6 if dtype == torch.float32 else 0, atol=0)
    e, grad, atomic = _run_oracle(g, hypers, params, dtype)
    np.testing.assert_allclose(e.numpy(), g[f"energies_{sfx}"], rtol=tol, atol=tol)
    np.testing.assert_allclose(atomic.detach().numpy(), g[f"atomic_{sfx}"], rtol=tol, atol=tol)
    scale = np.abs(g[f"grad_{sfx}"]).max()
    assert np.abs(grad.numpy() - g[f"grad_{sfx}"]).max() / scale < tol


@pytest.mark.parametrize("dtype,sfx,tol", [(torch.float64, "f64", 1e-10), (torch.float32, "f32", 1e-5)])
def test_oracle_default_box64(golden_dir, dtype, sfx, tol):
    g = _load(golden_dir, "pet_default_box64.npz")
    hypers = dict(opet.DEFAULT_HYPERS)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, dtype)
    e, grad, atomic = _run_oracle(g, hypers, params, dtype)
    np.testing.assert_allclose(e.numpy(), g[f"energies_{sfx}"], rtol=tol, atol=tol)
    np.testing.assert_allclose(atomic.detach().numpy(), g[f"atomic_{sfx}"], rtol=tol, atol=10 * tol)
    scale = np.abs(g[f"grad_{sfx}"]).max()
    assert np.abs(grad.numpy() - g[f"grad_{sfx}"]).max() / scale < tol


@pytest.mark.parametrize("dtype,sfx,tol", [(torch.float64, "f64", 1e-10), (torch.float32, "f32", 1e-5)])
def test_oracle_silu_activation_box64(golden_dir, dtype, sfx, tol):
    """activation = "SiLU" (transformer.py:32-49) against the reference's own output (make_golden.py --silu)."""
    g = _load(golden_dir, "pet_silu_box64.npz")
    hypers = dict(opet.DEFAULT_HYPERS, activation="SiLU")
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, dtype)
    assert params["gnn_layers.0.trans.layers.0.mlp.w_in.weight"].shape == (hypers["d_feedforward"], hypers["d_pet"])
    e, grad, atomic = _run_oracle(g, hypers, params, dtype)
    np.testing.assert_allclose(e.numpy(), g[f"energies_{sfx}"], rtol=tol, atol=tol)
    np.testing.assert_allclose(atomic.detach().numpy(), g[f"atomic_{sfx}"], rtol=tol, atol=10 * tol)
    scale = np.abs(g[f"grad_{sfx}"]).max()
    assert np.abs(grad.numpy() - g[f"grad_{sfx}"]).max() / scale < tol


def test_nl_oracle_tree_vs_bruteforce():
    """The two statements of the NL contract agree (sorted (i,j,S) sets), including
    self-images in a cell smaller than the cutoff and a triclinic cell."""
    rng = np.random.default_rng(0)
    cases = [
        (np.array([[0.0, 0, 0], [1.5, 1.5, 1.5]]), 3.5 * np.eye(3), [True] * 3),
        (rng.uniform(-3, 9, (30, 3)), np.array([[6.0, 0, 0], [1.5, 5.5, 0], [-1.0, 0.7, 7.0]]), [True] * 3),
        (rng.uniform(0, 6, (25, 3)), np.array([[6.0, 0, 0], [0, 6.0, 0], [0, 0, 6.0]]), [True, True, False]),
        (rng.uniform(0, 7, (20, 3)), np.zeros((3, 3)), [False] * 3),
        # ONE periodic direction (wires, chains): two rows of the effective cell have to be completed
        (rng.uniform(-2, 8, (30, 3)), np.array([[5.0, 0.4, 0], [0.3, 6.0, 0.2], [0, 0.5, 7.0]]), [True, False, False]),
        (rng.uniform(-2, 8, (30, 3)), np.array([[5.0, 0.4, 0], [0.3, 6.0, 0.2], [0, 0.5, 7.0]]), [False, True, False]),
        (rng.uniform(-2, 8, (30, 3)), np.array([[5.0, 0.4, 0], [0.3, 6.0, 0.2], [0, 0.5, 3.0]]), [False, False, True]),
    ]
    for pos, cell, pbc in cases:
        a = onl.neighbor_list(pos, cell, pbc, 4.5)
        b = onl.neighbor_list_bruteforce(pos, cell, pbc, 4.5)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(a[2], b[2])
        np.testing.assert_allclose(a[3], b[3], atol=1e-12)
        # full list: every (i,j,S) has its (j,i,-S)
        fwd = set(map(tuple, np.column_stack([a[0], a[1], a[2]]).tolist()))
        assert all((j, i, -x, -y, -z) in fwd for (i, j, x, y, z) in fwd)


def test_product_generators_equal_oracle_generators():
    """metatrain_amd.synthetic / metatrain_amd.pet.hypers carry their own copies of the synthetic
    input + weight generators (the product may not import the oracle): keep them identical."""
    from metatrain_amd import synthetic
    from metatrain_amd.pet import default_hypers

    assert default_hypers() == opet.DEFAULT_HYPERS
    a = synthetic.synthetic_params(opet.DEFAULT_HYPERS, [1, 6, 7, 8], {"energy": 1}, 3)
    b = opet.synthetic_params(opet.DEFAULT_HYPERS, [1, 6, 7, 8], {"energy": 1}, 3)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    for x, y in zip(synthetic.random_box(50, 9), opet.random_box(50, 9)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("method", ["solver", "grid"])
@pytest.mark.parametrize("case", ["box64", "two_systems"])
def test_oracle_adaptive_cutoff_matches_reference(golden_dir, case, method):
    """num_neighbors_adaptive -- the oracle's restatement of adaptive_cutoff.py:110-229 ("solver") and
    :232-395 ("grid", the legacy method older checkpoints keep) + structures.py:225-263 against fixtures generated from
    the reference."""
    hypers = dict(opet.DEFAULT_HYPERS, num_neighbors_adaptive=12, adaptive_cutoff_method=method,
                  cutoff_width_adaptive=1.0)
    name = "adaptive" if method == "solver" else "adaptive_grid"
    g = dict(np.load(os.path.join(golden_dir, f"pet_{name}_{case}.npz")))
    b = dict(np.load(os.path.join(golden_dir, f"batch_{name}_{case}.npz")))
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    table = torch.full((9,), -1, dtype=torch.long)
    table[torch.tensor([1, 6, 7, 8])] = torch.arange(4)
    out = opet.batch_tensors(hypers, table, t("in_positions").float(), t("in_cells").float(), t("in_centers"),
                             t("in_neighbors"), t("in_cell_shifts"), t("in_species"), t("in_system_indices"))
    for k in ("padding_mask", "reverse_neighbor_index", "centers", "neighbors", "cell_shifts",
              "element_indices_neighbors", "nef_to_edges_neighbor"):
        assert np.array_equal(out[k], b[k]), k
    np.testing.assert_allclose(out["atomic_cutoffs_stats"], b["atomic_cutoffs_stats"], rtol=2e-6)
    np.testing.assert_allclose(out["cutoff_factors"], b["cutoff_factors"], rtol=1e-4, atol=2e-6)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    e, grad, atomic = opet.energy_and_gradient(
        params, hypers, t("in_positions"), t("in_cells"), t("in_centers"), t("in_neighbors"), t("in_cell_shifts"),
        t("in_species"), t("in_system_indices"))
    np.testing.assert_allclose(e.numpy(), g["energies_f64"], rtol=1e-10)
    np.testing.assert_allclose(atomic.numpy(), g["atomic_f64"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(grad.numpy(), g["grad_f64"], rtol=1e-8, atol=1e-11)


def test_oracle_cosine_cutoff_matches_reference(golden_dir):
    """cutoff_function = "Cosine" (pet/modules/utilities.py:25-39): batch_data and E / dE/dR of the reference."""
    hypers = dict(opet.DEFAULT_HYPERS, cutoff_function="Cosine")
    b = _load(golden_dir, "batch_cosine_box64.npz")
    table = torch.full((9,), -1, dtype=torch.long)
    table[torch.tensor([1, 6, 7, 8])] = torch.arange(4)
    t = lambda d, k: torch.tensor(d[k])  # noqa: E731
    out = opet.batch_tensors(hypers, table, t(b, "in_positions"), t(b, "in_cells"), t(b, "in_centers"),
                             t(b, "in_neighbors"), t(b, "in_cell_shifts"), t(b, "in_species"), t(b, "in_system_indices"))
    for k in INT_KEYS:
        assert np.array_equal(out[k], b[k]), k
    np.testing.assert_allclose(out["cutoff_factors"], b["cutoff_factors"], rtol=2e-6, atol=2e-6)
    g = _load(golden_dir, "pet_cosine_box64.npz")
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    e, grad, atomic = _run_oracle(g, hypers, params, torch.float64)
    np.testing.assert_allclose(e.numpy(), g["energies_f64"], rtol=1e-10)
    np.testing.assert_allclose(atomic.numpy(), g["atomic_f64"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(grad.numpy(), g["grad_f64"], rtol=1e-8, atol=1e-11)


def test_oracle_training_step_matches_reference_fixture(golden_dir):
    """SURVEY §8(c) fixture (iv): the oracle's double backward (loss on E / atom and on dE/dR) against what the reference
    produced in train() mode with manual attention (make_golden.py --train): loss terms, every parameter's gradient
    norm, four full gradient tensors -- fp64, 1e-9."""
    g = _load(golden_dir, "pet_train_two_systems.npz")
    hypers = dict(opet.DEFAULT_HYPERS)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    p64 = {k: (v if k == "species_to_species_index" else v.clone().requires_grad_(True)) for k, v in params.items()}
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    pos = t("in_positions").clone().requires_grad_(True)
    s = t("in_system_indices")
    atomic = opet.pet_atomic_energies(p64, hypers, pos, t("in_cells"), t("in_centers"), t("in_neighbors"),
                                      t("in_cell_shifts"), t("in_species"), s)[:, 0]
    e = torch.zeros(2, dtype=torch.float64).index_add(0, s, atomic)
    (grad,) = torch.autograd.grad(e.sum(), pos, create_graph=True)
    n_atoms = torch.tensor([64.0, 40.0], dtype=torch.float64)
    loss_e = (((e - t("e_target")) / n_atoms) ** 2).mean()
    loss_f = ((grad - t("g_target")) ** 2).mean()
    (loss_e + loss_f).backward()
    np.testing.assert_allclose([float(loss_e + loss_f), float(loss_e), float(loss_f)], g["loss_f64"], rtol=1e-10)
    keys = [str(k) for k in g["param_keys"]]
    assert keys == [k for k in p64 if k != "species_to_species_index"]
    norms = np.array([float(p64[k].grad.norm()) for k in keys])
    np.testing.assert_allclose(norms, g["grad_norms_f64"], rtol=1e-8, atol=1e-12)
    for k in [k[len("dparam_f64::"):] for k in g if k.startswith("dparam_f64::")]:
        ref = g["dparam_f64::" + k]
        assert np.abs(p64[k].grad.numpy() - ref).max() < 1e-9 * np.abs(ref).max(), k
    # the reference's own fp32 training pass sits this far from its fp64 one (context for the GPU tolerance)
    assert np.abs(g["grad_norms_f32"] - g["grad_norms_f64"]).max() < 1e-4 * g["grad_norms_f64"].max()


def test_box10000_golden_is_consistent(golden_dir):
    """The 10 000-atom fixture (BASELINE.json's metric size): the fp64 oracle's per-atom energies equal the
    reference's fp64 ones to 1e-12, its gradient equals the reference's fp32 gradient to the fp32 noise floor, the
    stored inputs are the documented generator's output."""
    g = _load(golden_dir, "pet_default_box10000.npz")
    pos, z, cell = opet.random_box(10000, seed=0)
    assert np.array_equal(pos.numpy(), g["in_positions"]) and np.array_equal(z.numpy(), g["in_species"])
    assert np.array_equal(cell.numpy(), g["in_cell"])
    scale = np.abs(g["atomic_ref_f64"]).max()
    assert np.abs(g["atomic_f64"] - g["atomic_ref_f64"]).max() < 1e-12 * scale
    assert np.abs(g["atomic_ref_f32"] - g["atomic_ref_f64"]).max() < 1e-5 * scale
    assert np.abs(g["grad_f64"] - g["grad_ref_f32"]).max() < 1e-5 * np.abs(g["grad_f64"]).max()
    assert abs(g["grad_f64"].sum(0)).max() < 1e-9  # Newton's third law in fp64


@pytest.mark.parametrize("tag", ["legacy", "layernorm", "postln", "residual"])
def test_oracle_variants_match_reference(golden_dir, tag):
    """LayerNorm normalisation (transformer.py:176-186), PostLN layers (:236-262), the residual
    featuriser (backend.py:589-649) -- alone and combined with activation = "SiLU" as the reference's checkpoint
    upgrade leaves older models (pet/checkpoints.py:190-205, "legacy") -- against make_golden.py --variants."""
    delta = {"legacy": dict(normalization="LayerNorm", activation="SiLU", transformer_type="PostLN", featurizer_type="residual"),
             "layernorm": dict(normalization="LayerNorm"), "postln": dict(transformer_type="PostLN"),
             "residual": dict(featurizer_type="residual")}[tag]
    hypers = dict(opet.DEFAULT_HYPERS, **delta)
    g = _load(golden_dir, f"pet_variant_{tag}_box64.npz")
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    e, grad, atomic = _run_oracle(g, hypers, params, torch.float64)
    np.testing.assert_allclose(e.numpy(), g["energies_f64"], rtol=1e-10)
    np.testing.assert_allclose(atomic.numpy(), g["atomic_f64"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(grad.numpy(), g["grad_f64"], rtol=1e-8, atol=1e-11)
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    _, nfs, efs, graph = opet.pet_atomic_energies(params, hypers, t("in_positions"), t("in_cells"), t("in_centers"),
                                                  t("in_neighbors"), t("in_cell_shifts").long(), t("in_species"),
                                                  t("in_system_indices"), return_features="all")
    assert len(nfs) == int(g["n_readout"])
    for l, nf in enumerate(nfs):
        np.testing.assert_allclose(nf.numpy(), g[f"node_features_{l}_f64"], rtol=1e-8, atol=1e-11)


SIZES = {   # = tests/golden/make_golden.py::SIZES
    "s64": dict(d_pet=64, d_node=128, d_feedforward=128, d_head=64, num_heads=4),
    "flat32": dict(d_pet=32, d_node=32, d_feedforward=48, d_head=24, num_heads=2),
    "flat32_legacy": dict(d_pet=32, d_node=32, d_feedforward=48, d_head=24, num_heads=2, normalization="LayerNorm",
                          activation="SiLU", transformer_type="PostLN", featurizer_type="residual"),
    "wide256": dict(d_pet=256, d_node=512, d_feedforward=320, d_head=96, num_heads=4),
    "minimal": dict(d_pet=1, d_node=1, d_feedforward=1, d_head=1, num_heads=1, num_attention_layers=1, num_gnn_layers=1),
}


@pytest.mark.parametrize("tag", list(SIZES))
def test_oracle_other_model_sizes_match_reference(golden_dir, tag):
    """The reference is size-generic (pet/documentation.py:196-213); with d_node == d_pet it holds Identity modules in
    place of centre contraction / expansion / MLP (transformer.py:189-201), and its architecture suites run at
    d_pet = 1 (pet/tests/test_basic.py:22-32) -- against make_golden.py --sizes."""
    hypers = dict(opet.DEFAULT_HYPERS, **SIZES[tag])
    g = _load(golden_dir, f"pet_size_{tag}_box64.npz")
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    if hypers["d_node"] == hypers["d_pet"]:
        assert not any("center_" in k for k in params)
    e, grad, atomic = _run_oracle(g, hypers, params, torch.float64)
    np.testing.assert_allclose(atomic.numpy(), g["atomic_f64"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(grad.numpy(), g["grad_f64"], rtol=1e-8, atol=1e-11)
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    _, nfs, _, _ = opet.pet_atomic_energies(params, hypers, t("in_positions"), t("in_cells"), t("in_centers"),
                                            t("in_neighbors"), t("in_cell_shifts").long(), t("in_species"),
                                            t("in_system_indices"), return_features="all")
    assert len(nfs) == int(g["n_readout"])
    for l, nf in enumerate(nfs):
        np.testing.assert_allclose(nf.numpy(), g[f"node_features_{l}_f64"], rtol=1e-8, atol=1e-11)
    # the product's own generator knows the schema of these sizes too
    from metatrain_amd import synthetic

    a = synthetic.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0)
    b = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0)
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("tag", ["feedforward", "residual"])
def test_oracle_system_conditioning_matches_reference(golden_dir, tag):
    """system_conditioning = True (conditioning.py; backend.py:121-130, 517-545, 607-630): charge / spin-multiplicity
    embedding added to the node features leaving every GNN layer -- two systems with different charge and multiplicity,
    against make_golden.py --conditioning."""
    hypers = dict(opet.DEFAULT_HYPERS, system_conditioning=True,
                  featurizer_type="residual" if tag == "residual" else "feedforward")
    g = _load(golden_dir, f"pet_conditioning_{tag}.npz")
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    e, grad, atomic = opet.energy_and_gradient(
        params, hypers, t("in_positions"), t("in_cells"), t("in_centers"), t("in_neighbors"), t("in_cell_shifts").long(),
        t("in_species"), t("in_system_indices"), charge=t("in_charge"), spin_multiplicity=t("in_spin_multiplicity"))
    np.testing.assert_allclose(atomic.numpy(), g["atomic_f64"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(grad.numpy(), g["grad_f64"], rtol=1e-8, atol=1e-11)
    _, nfs, _, _ = opet.pet_atomic_energies(
        params, hypers, t("in_positions"), t("in_cells"), t("in_centers"), t("in_neighbors"), t("in_cell_shifts").long(),
        t("in_species"), t("in_system_indices"), return_features="all", charge=t("in_charge"),
        spin_multiplicity=t("in_spin_multiplicity"))
    assert len(nfs) == int(g["n_readout"])
    for l, nf in enumerate(nfs):
        np.testing.assert_allclose(nf.numpy(), g[f"node_features_{l}_f64"], rtol=1e-8, atol=1e-11)
    # the conditioning really acts: without it the energies differ
    e0, _, _ = opet.energy_and_gradient(
        params, hypers, t("in_positions"), t("in_cells"), t("in_centers"), t("in_neighbors"), t("in_cell_shifts").long(),
        t("in_species"), t("in_system_indices"))
    assert float((e0 - e).abs().max()) > 1e-3


def test_oracle_reproduces_the_reference_tests_hand_tables_of_the_grid_method(golden_dir):
    """The two known-answer tables of the reference's own adaptive-cutoff tests (pet/tests/test_adaptive_cutoff.py:
    232-292: smoothed neighbour counts on a probe grid, Gaussian probe weights), torch.allclose as there."""
    import json

    kat = json.load(open(os.path.join(golden_dir, "reference_known_answers.json")))
    a = kat["effective_num_neighbors"]
    n_eff = opet.grid_effective_num_neighbors(torch.tensor(a["edge_distances"]), torch.tensor(a["probe_cutoffs"]),
                                              torch.tensor(a["centers"]), a["num_nodes"], a["width"])
    assert torch.allclose(n_eff, torch.tensor(a["expected"]))
    b = kat["gaussian_cutoff_weights"]
    w = opet.grid_gaussian_weights(torch.tensor(b["effective_num_neighbors"]), b["num_neighbors_adaptive"])
    assert torch.allclose(w, torch.tensor(b["expected"]))


def test_energy_reads_one_cutoff_further_than_the_declared_interaction_range(golden_dir):
    """``tests/golden/check_interaction_range.py`` ran the imported REFERENCE on a four-atom chain A-B-C-D (4 A spacing,
    4.5 A cutoff, two GNN layers): dE_A/dR_D is not zero although D is 12 A from A, beyond the ``num_gnn_layers x cutoff``
    = 9 A of ``pet/model.py:1004`` (the reversed-edge term of the last combination, ``backend.py:559-575``, reads one hop
    further). The oracle reproduces the reference's numbers; ``metatrain_amd/pet/partition.py`` sizes its halos by it."""
    import json

    ref = json.load(open(os.path.join(golden_dir, "reference_interaction_range.json")))
    hyp = dict(opet.DEFAULT_HYPERS)
    params = opet.synthetic_params(hyp, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    pos = torch.tensor([[0.0, 0, 0], [4.0, 0.3, 0], [8.0, -0.2, 0.4], [12.0, 0.1, -0.3]], dtype=torch.float64,
                       requires_grad=True)
    z = torch.tensor([6, 1, 8, 7])
    cell = torch.zeros(1, 3, 3, dtype=torch.float64)
    i, j, s, _ = onl.neighbor_list(pos.detach().numpy(), cell[0].numpy(), [False] * 3, hyp["cutoff"])
    atomic = opet.pet_atomic_energies(params, hyp, pos, cell, torch.tensor(i), torch.tensor(j), torch.tensor(s),
                                      z, torch.zeros(4, dtype=torch.long))
    (g,) = torch.autograd.grad(atomic[0, 0], pos)
    assert ref["distance_A_to_D"] > ref["declared_interaction_range_A"] + 2.9
    for n, k in enumerate("ABCD"):
        np.testing.assert_allclose(g[n].numpy(), ref["dE_A_dR"][k], rtol=1e-9, atol=1e-14)
    assert float(g[3].abs().max()) > 1e-3 * float(g[1].abs().max())  # far above rounding: a real dependence


def test_oracle_matches_reference_on_several_targets_blocks_and_properties(golden_dir):
    """``make_golden.py --multitarget``: the reference's predictions for an energy, a target with two blocks (3 and 6
    properties) and the non-conservative stress (symmetrised, divided by the volume: backend.py:780-813), and dE/dR of a
    weighted sum of them -- the oracle's per-block restatement against it."""
    g = dict(np.load(os.path.join(golden_dir, "pet_multitarget_box50.npz")))
    hyp = dict(opet.DEFAULT_HYPERS)
    targets = {"energy": 1, "multi": {"a": 3, "b": 6}, "non_conservative_stress": 9}
    p64 = opet.synthetic_params(hyp, [1, 6, 7, 8], targets, 0, torch.float64)
    r = torch.tensor(g["in_positions"]).requires_grad_(True)
    cells = torch.tensor(g["in_cells"])
    args = (hyp, r, cells, torch.tensor(g["in_centers"]), torch.tensor(g["in_neighbors"]),
            torch.tensor(g["in_cell_shifts"]), torch.tensor(g["in_species"]), torch.tensor(g["in_system_indices"]))
    ra = opet.pet_atomic_energies(p64, *args, "multi", "a")
    rb = opet.pet_atomic_energies(p64, *args, "multi", "b")
    re = opet.pet_atomic_energies(p64, *args, "energy")
    rs = opet.pet_atomic_energies(p64, *args, "non_conservative_stress")
    (gr,) = torch.autograd.grad((ra * torch.tensor(g["wa"])).sum() + (rb * torch.tensor(g["wb"])).sum() + re.sum(), r)
    np.testing.assert_allclose(re.detach().numpy(), g["energy"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ra.detach().numpy(), g["multi_a"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(rb.detach().numpy(), g["multi_b"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gr.numpy(), g["grad"], rtol=1e-9, atol=1e-12)
    t = rs.detach().reshape(-1, 3, 3, 1) / float(torch.det(cells[0]).abs())
    np.testing.assert_allclose(((t + t.transpose(1, 2)) / 2).numpy(), g["non_conservative_stress"], rtol=1e-10, atol=1e-14)

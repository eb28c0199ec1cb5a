"""GPU tests of the drop-in boundary, written after the reference's own
pet/tests/test_backend.py and pet/tests/test_regression.py: the backend consumes and returns
plain tensors, and energy / forces / strain gradient come out of torch.autograd.grad."""
import os

import numpy as np
import pytest
import torch

from oracle import nl as onl
from oracle import pet as opet

pytestmark = pytest.mark.gpu
TOL = 1e-5


def relmax(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / np.abs(b).max())


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _backend(dev, target="energy", seed=None):
    from metatrain_amd.pet import PETBackend, default_hypers

    hypers = default_hypers()
    if seed is not None:
        torch.manual_seed(seed)
    be = PETBackend(hypers, [1, 6, 7, 8])
    be.add_output(target, {target: [1]})
    if seed is None:
        be.load_state_dict(opet.synthetic_params(hypers, [1, 6, 7, 8], {target: 1}))
    return be.to(dev).eval(), hypers


def _inputs(pos, z, cell, pbc, cutoff, dev):
    i, j, s, _ = onl.neighbor_list(np.asarray(pos, dtype=np.float64), np.asarray(cell, dtype=np.float64), pbc, cutoff)
    return (torch.tensor(i, device=dev), torch.tensor(j, device=dev), torch.tensor(s, device=dev),
            torch.tensor(z, device=dev), torch.zeros(len(z), dtype=torch.long, device=dev))


def test_backend_runs_on_plain_tensors(dev):
    """pet/tests/test_backend.py:122-150 (water molecule, non-periodic)."""
    be, hypers = _backend(dev)
    pos = torch.tensor([[0.0, 0.0, 0.119], [0.0, 0.757, -0.477], [0.0, -0.757, -0.477]], device=dev)
    cells = torch.zeros(1, 3, 3, device=dev)
    i, j, s, z, sysidx = _inputs(pos.cpu().numpy(), [8, 1, 1], np.zeros((3, 3)), [False] * 3, hypers["cutoff"], dev)
    batch = be.preprocess(pos, i, j, z, cells, s, sysidx, 1.0)
    assert isinstance(batch, dict) and len(batch) == 12
    assert all(isinstance(v, torch.Tensor) for v in batch.values())
    nodes, edges = be.calculate_features(batch)
    assert nodes[0].shape == (3, 256) and edges[0].shape == (3, 2, 128)
    pred, node_ll, edge_ll = be.predict(nodes, edges, batch, cells, sysidx, ["energy"])
    assert pred["energy"][0].shape == (3, 1)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    ref = opet.pet_atomic_energies(params, hypers, pos.cpu().double(), cells.cpu().double(), i.cpu(), j.cpu(),
                                   s.cpu().long(), z.cpu(), sysidx.cpu())
    assert (pred["energy"][0].cpu().double() - ref).abs().max() / ref.abs().max() < TOL
    # the per-atom "feature" and last-layer-feature outputs of the model wrapper (pet/model.py:730-875)
    feat, llf = be.auxiliary_outputs(nodes, edges, batch, "energy")
    _, feat64, llf64 = opet.pet_atomic_energies(params, hypers, pos.cpu().double(), cells.cpu().double(), i.cpu(),
                                                j.cpu(), s.cpu().long(), z.cpu(), sysidx.cpu(), return_aux=True)
    assert feat.shape == (3, 384) and llf.shape == (3, 256)
    assert relmax(feat.cpu().numpy(), feat64.numpy()) < TOL and relmax(llf.cpu().numpy(), llf64.numpy()) < TOL


def test_energy_forces_and_strain_gradient_via_autograd(dev):
    """pet/tests/test_backend.py:69-117: 2-atom 3.5 A cubic C/O cell with a 4.5 A cutoff (periodic
    self images), energy / forces / strain gradient from torch.autograd.grad with the strain trick."""
    be, hypers = _backend(dev)
    base = torch.tensor([[0.0, 0.0, 0.0], [1.5, 1.5, 1.5]])
    cell = 3.5 * torch.eye(3)
    i, j, s, z, sysidx = _inputs(base.numpy(), [6, 8], cell.numpy(), [True] * 3, hypers["cutoff"], dev)

    def energy(pos_leaf, strain, backend_like):
        return backend_like(pos_leaf @ strain, (cell.to(pos_leaf) @ strain)[None])

    def hip(pos, cells):
        batch = be.preprocess(pos, i, j, z, cells, s, sysidx, 1.0)
        nodes, edges = be.calculate_features(batch)
        pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy"])
        return pred["energy"][0].sum()

    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)

    def oracle(pos, cells):
        return opet.pet_atomic_energies(params, hypers, pos, cells, i.cpu(), j.cpu(), s.cpu().long(), z.cpu(),
                                        sysidx.cpu()).sum()

    p = base.to(dev).requires_grad_(True)
    st = torch.eye(3, device=dev).requires_grad_(True)
    e = energy(p, st, hip)
    gp, gs = torch.autograd.grad(e, [p, st])
    p64 = base.double().requires_grad_(True)
    st64 = torch.eye(3, dtype=torch.float64).requires_grad_(True)
    e64 = energy(p64, st64, oracle)
    gp64, gs64 = torch.autograd.grad(e64, [p64, st64])
    assert abs(float(e) - float(e64)) / abs(float(e64)) < TOL
    assert (gp.cpu().double() - gp64).abs().max() / gp64.abs().max() < TOL
    assert (gs.cpu().double() - gs64).abs().max() / gs64.abs().max() < TOL


@pytest.mark.parametrize("variant", ["default", "small", "adaptive_solver", "adaptive_grid", "small_adaptive"])
def test_empty_isolated_and_dissociated_systems_through_the_mirror(dev, variant):
    """pet/tests/test_functionality.py:79-159 and test_adaptive_cutoff.py:408-500: an EMPTY system (zero-sized outputs, and a
    backward that returns a [0, 3] gradient), one isolated atom, two atoms 100 A apart and a bonded pair -- tuned and
    size-generic kernels, fixed and adaptive cutoffs (both methods): finite, equal to the oracle."""
    from metatrain_amd.pet import PETBackend

    extra = {"default": {}, "small": dict(d_pet=8, d_head=8, d_node=16, d_feedforward=8, num_heads=2),
             "adaptive_solver": dict(num_neighbors_adaptive=5.0),
             "adaptive_grid": dict(num_neighbors_adaptive=5.0, adaptive_cutoff_method="grid"),
             "small_adaptive": dict(d_pet=8, d_head=8, d_node=16, d_feedforward=8, num_heads=2, num_neighbors_adaptive=5.0)}[variant]
    hypers = dict(opet.DEFAULT_HYPERS, **extra)
    types = [1, 6, 7, 8]
    params = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float32)
    be = PETBackend(hypers, types)
    be.add_output("energy", {"energy": [1]})
    be.load_state_dict(params, strict=True)
    be = be.to(dev).eval()
    p64 = {k: (v if k == "species_to_species_index" else v.double()) for k, v in params.items()}
    systems = {"empty": (np.zeros((0, 3)), []), "isolated": (np.zeros((1, 3)), [6]),
               "dissociated": (np.array([[0, 0, 0], [0, 0, 100.0]]), [6, 6]), "pair": (np.array([[0, 0, 0], [0, 0, 1.2]]), [6, 8])}
    for name, (pos, z) in systems.items():
        i, j, s, _ = onl.neighbor_list(pos, np.zeros((3, 3)), [False] * 3, hypers["cutoff"])
        ti = lambda a: torch.tensor(np.asarray(a), dtype=torch.long, device=dev)  # noqa: E731
        p = torch.tensor(pos, dtype=torch.float32, device=dev).reshape(-1, 3).requires_grad_(True)
        cells = torch.zeros(1, 3, 3, device=dev)
        sysidx = torch.zeros(len(z), dtype=torch.long, device=dev)
        batch = be.preprocess(p, ti(i), ti(j), ti(z), cells, ti(s).reshape(-1, 3), sysidx, float(hypers["cutoff_width_adaptive"]))
        nodes, edges = be.calculate_features(batch)
        pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy"])
        atomic = pred["energy"][0]
        (grad,) = torch.autograd.grad(atomic.sum(), p)
        assert atomic.shape == (len(z), 1) and grad.shape == (len(z), 3), name
        assert torch.isfinite(atomic).all() and torch.isfinite(grad).all(), name
        if len(z):
            ref = opet.pet_atomic_energies(p64, hypers, torch.tensor(pos).double().reshape(-1, 3),
                                           torch.zeros(1, 3, 3, dtype=torch.float64), torch.tensor(i).long(),
                                           torch.tensor(j).long(), torch.tensor(s).long().reshape(-1, 3), torch.tensor(z),
                                           torch.zeros(len(z), dtype=torch.long))
            assert float((atomic.detach().cpu().double() - ref).abs().max() / ref.abs().max()) < TOL, name


@pytest.mark.parametrize("fullgraph", [False, True])
def test_backend_torch_compile(dev, fullgraph):
    """pet/tests/test_backend.py:177-330: ``torch.compile`` of the three backend calls (under the Dynamo flags the reference
    sets) matches eager execution on the adaptive-cutoff path and a periodic system: ``preprocess``'s twelve tensors,
    energy, forces and the strain gradient -- and the eager values are the oracle's."""
    from metatrain_amd.pet import PETBackend, default_hypers

    hypers = dict(default_hypers(), num_neighbors_adaptive=5.0)

    def make():
        be = PETBackend(hypers, [1, 6, 7, 8])
        be.add_output("energy", {"energy": [1]})
        be.load_state_dict(opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}))
        return be.to(dev).eval()

    base = torch.tensor([[0.0, 0.0, 0.0], [1.5, 1.5, 1.5]])
    cell = 3.5 * torch.eye(3)
    i, j, s, z, sysidx = _inputs(base.numpy(), [6, 8], cell.numpy(), [True] * 3, hypers["cutoff"], dev)
    cwa = float(hypers["cutoff_width_adaptive"]) if "cutoff_width_adaptive" in hypers else 1.0

    def results(be):
        p = base.to(dev).requires_grad_(True)
        st = torch.eye(3, device=dev).requires_grad_(True)
        cells = (cell.to(dev) @ st)[None]
        batch = be.preprocess(p @ st, i, j, z, cells, s, sysidx, cwa)
        nodes, edges = be.calculate_features(batch)
        pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy"])
        e = pred["energy"][0].sum()
        gp, gs = torch.autograd.grad(e, [p, st])
        return {k: v.detach() for k, v in batch.items()}, e.detach(), gp, gs

    batch_e, e_e, gp_e, gs_e = results(make())
    be = make()
    with torch._dynamo.config.patch(capture_scalar_outputs=True, capture_dynamic_output_shape_ops=True, specialize_int=True):
        be.preprocess = torch.compile(be.preprocess, fullgraph=fullgraph)
        be.calculate_features = torch.compile(be.calculate_features, fullgraph=fullgraph)
        be.predict = torch.compile(be.predict, fullgraph=fullgraph)
        batch_c, e_c, gp_c, gs_c = results(be)
    assert set(batch_c) == set(batch_e)
    for key in batch_e:
        assert batch_e[key].shape == batch_c[key].shape, key
        torch.testing.assert_close(batch_e[key], batch_c[key])
    torch.testing.assert_close(e_c, e_e)
    torch.testing.assert_close(gp_c, gp_e)
    torch.testing.assert_close(gs_c, gs_e)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    p64 = base.double().requires_grad_(True)
    st64 = torch.eye(3, dtype=torch.float64).requires_grad_(True)
    e64 = opet.pet_atomic_energies(params, hypers, p64 @ st64, (cell.double() @ st64)[None], i.cpu(), j.cpu(), s.cpu().long(),
                                   z.cpu(), sysidx.cpu()).sum()
    gp64, gs64 = torch.autograd.grad(e64, [p64, st64])
    assert abs(float(e_c) - float(e64)) / abs(float(e64)) < TOL
    assert (gp_c.cpu().double() - gp64).abs().max() / gp64.abs().max() < TOL
    assert (gs_c.cpu().double() - gs64).abs().max() / gs64.abs().max() < TOL


def test_seed0_backend_reproduces_reference_regression_energies(dev, golden_dir):
    """The reference's own golden numbers (pet/tests/test_regression.py:66-74): fresh model under
    torch.manual_seed(0), first five QM9 frames, per-system energies."""
    be, hypers = _backend(dev, target="mtt::U0", seed=0)
    g = dict(np.load(os.path.join(golden_dir, "qm9_first5.npz")))
    got = []
    for k in range(5):
        zk, xyz = g[f"z{k}"], g[f"pos{k}"]
        i, j, s, z, sysidx = _inputs(xyz, zk, np.zeros((3, 3)), [False] * 3, hypers["cutoff"], dev)
        pos = torch.tensor(xyz, dtype=torch.float32, device=dev)
        cells = torch.zeros(1, 3, 3, device=dev)
        batch = be.preprocess(pos, i, j, z, cells, s, sysidx, 1.0)
        nodes, edges = be.calculate_features(batch)
        pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["mtt::U0"])
        got.append(float(pred["mtt::U0"][0].sum()))
    # torch.testing.assert_close fp32 defaults, the tolerance the reference applies to itself
    np.testing.assert_allclose(got, g["expected_reference_test"], rtol=1.3e-6, atol=1e-5)


def test_parameter_update_is_picked_up(dev):
    """Weights changed in place (optimizer step / load_state_dict) must reach the packed copy."""
    be, hypers = _backend(dev)
    pos, z, cell = opet.random_box(40, 3)
    i, j, s, zz, sysidx = _inputs(pos.numpy(), z.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"], dev)

    def run():
        batch = be.preprocess(pos.to(dev), i, j, zz, cell[None].to(dev), s, sysidx, 1.0)
        n, e = be.calculate_features(batch)
        return be.predict(n, e, batch, cell[None].to(dev), sysidx, ["energy"])[0]["energy"][0].sum().item()

    e0 = run()
    with torch.no_grad():
        be.node_last_layers["energy"][0]["energy"].bias.add_(1.0)
    assert abs(run() - (e0 + 40.0)) < 1e-3


@pytest.mark.parametrize("activation,conditioned,normalization", [
    ("SwiGLU", False, "RMSNorm"), ("SiLU", False, "RMSNorm"), ("SwiGLU", True, "RMSNorm"), ("SiLU", True, "LayerNorm")])
def test_training_through_the_mirror_fills_parameter_grads(golden_dir, activation, conditioned, normalization):
    """pet/trainer.py:417-462 through the torch mirror: autograd.grad(E, R, create_graph=True), a loss on
    energies and dE/dR, loss.backward() -> parameter.grad; against torch's double backward through the fp64
    oracle with the same weights. With activation = "SiLU" the w_in gradients are those of the tied projection; with
    system conditioning the charge / spin embeddings and their projection are trained too; with normalization =
    "LayerNorm" the norm biases are."""
    from metatrain_amd.pet import PETBackend, default_hypers

    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(golden_dir, "batch_two_systems.npz")))
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    hypers = dict(default_hypers(), activation=activation, system_conditioning=conditioned, normalization=normalization)
    charge, spin = torch.tensor([2, -1]), torch.tensor([3, 1])
    kw = dict(charge=charge, spin_multiplicity=spin) if conditioned else {}
    types = [1, 6, 7, 8]
    params = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float32)
    gen = torch.Generator().manual_seed(5)
    if normalization == "LayerNorm":  # norm parameters start at (1, 0): move them
        for k in params:
            if ".norm_" in k:
                params[k] = params[k] + 0.3 * torch.randn(params[k].shape, generator=gen)
    be = PETBackend(hypers, types)
    be.add_output("energy", {"energy": [1]})
    be.load_state_dict(params, strict=True)
    be = be.to(dev)
    pos = t("in_positions").float().to(dev).requires_grad_(True)
    cells, sysidx = t("in_cells").float().to(dev), t("in_system_indices").to(dev)
    n = pos.shape[0]
    w_e = torch.rand(n, generator=gen) - 0.5
    tgt_g = 0.3 * torch.randn(n, 3, generator=gen)
    batch = be.preprocess(pos, t("in_centers").to(dev), t("in_neighbors").to(dev), t("in_species").to(dev), cells,
                          t("in_cell_shifts").to(dev), sysidx, 1.0)
    if conditioned:
        batch["charge"], batch["spin_multiplicity"], batch["system_indices"] = charge.to(dev), spin.to(dev), sysidx
    nf, ef = be.calculate_features(batch)
    pred, _, _ = be.predict(nf, ef, batch, cells, sysidx, ["energy"])
    atomic = pred["energy"][0][:, 0]
    (grad,) = torch.autograd.grad(atomic.sum(), pos, create_graph=True)
    loss = (w_e.to(dev) * atomic).sum() + ((grad - tgt_g.to(dev)) ** 2).sum()
    loss.backward()

    p64 = {k: (v if k == "species_to_species_index" else v.double().clone().requires_grad_(True))
           for k, v in params.items()}
    rpos = t("in_positions").double().clone().requires_grad_(True)
    a_ref = opet.pet_atomic_energies(p64, hypers, rpos, t("in_cells").double(), t("in_centers"), t("in_neighbors"),
                                     t("in_cell_shifts"), t("in_species"), t("in_system_indices").long(), "energy",
                                     **kw)[:, 0]
    (g_ref,) = torch.autograd.grad(a_ref.sum(), rpos, create_graph=True)
    l_ref = (w_e.double() * a_ref).sum() + ((g_ref - tgt_g.double()) ** 2).sum()
    keys = [k for k in p64 if k != "species_to_species_index"]
    ref = dict(zip(keys, torch.autograd.grad(l_ref, [p64[k] for k in keys], allow_unused=True)))
    assert abs(float(loss) - float(l_ref)) / abs(float(l_ref)) < 1e-5
    named = dict(be.named_parameters())
    worst = 0.0
    for k in keys:
        r = ref[k]
        got = named[k].grad
        if r is None:
            continue
        assert got is not None, k
        scale = float(r.abs().max())
        if scale > 1e-12:
            worst = max(worst, float((got.cpu().double() - r).abs().max()) / scale)
    assert worst < 1e-5, worst
    if conditioned:
        assert all(float(named[k].grad.abs().max()) > 0 for k in keys if k.startswith("system_conditioning."))


@pytest.mark.parametrize("activation,fixture", [("SwiGLU", "pet_default_box64.npz"), ("SiLU", "pet_silu_box64.npz")])
def test_torchscript_module_energy_and_forces(golden_dir, activation, fixture):
    """the TorchScript custom class (csrc/torch_ops.cpp), scripted, saved, re-loaded, then run:
    energies and forces (autograd inside TorchScript) against the golden fp64 reference values."""
    import io

    from metatrain_amd.pet import default_hypers, script

    dev = torch.device("cuda:0")
    hypers = dict(default_hypers(), activation=activation)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
    core = script.make_core(hypers, [1, 6, 7, 8], params, "energy")
    buf = io.BytesIO()
    torch.jit.save(torch.jit.script(script.EnergyAndForces(core)), buf)
    buf.seek(0)
    mod = torch.jit.load(buf)
    g = dict(np.load(os.path.join(golden_dir, fixture)))
    t = lambda k: torch.tensor(g[k]).to(dev)  # noqa: E731
    energies, forces = mod(t("in_positions").float(), t("in_cells").float(), t("in_centers"), t("in_neighbors"),
                           t("in_cell_shifts"), t("in_species"), t("in_system_indices"))
    e_ref, g_ref = g["energies_f64"].ravel(), g["grad_f64"]
    assert abs(float(energies[0]) - e_ref[0]) / abs(e_ref[0]) < TOL
    assert np.abs(-forces.cpu().numpy() - g_ref).max() / np.abs(g_ref).max() < TOL
    # second call on the same module (packed weights are cached on the device)
    e2, _ = mod(t("in_positions").float(), t("in_cells").float(), t("in_centers"), t("in_neighbors"),
                t("in_cell_shifts"), t("in_species"), t("in_system_indices"))
    assert torch.equal(e2, energies)


def test_exported_model_scaler_composition_selected_atoms_and_stress():
    """SURVEY §8(f)-2, the evaluation-time wrapper (pet/model.py:592-660): scaler factor, composition energies,
    selected_atoms, forces and the strain derivative, scripted, against the fp64 oracle with the strain trick."""
    from metatrain_amd.pet import default_hypers, script

    dev = torch.device("cuda:0")
    hypers = default_hypers()
    types = [1, 6, 7, 8]
    p32 = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float32)
    p64 = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float64)
    pos, z, cell = opet.random_box(48, seed=31)
    cell = cell.clone(); cell[1, 0] = 0.8; cell[2, 0] = -0.5
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"])
    i, j, s = torch.tensor(i), torch.tensor(j), torch.tensor(s)
    sysidx = torch.zeros(48, dtype=torch.int32)
    comp = torch.zeros(9); comp[[1, 6, 7, 8]] = torch.tensor([-0.5, -37.8, -54.6, -75.1])
    scale = 2.5
    keep = torch.arange(48) % 3 != 0
    mod = torch.jit.script(script.ExportedEnergyModel(script.make_core(hypers, types, p32, "energy"), scale, comp)).to(dev)
    args = (pos.to(dev), cell[None].to(dev), i.to(dev), j.to(dev), s.to(dev), z.to(dev), sysidx.to(dev))
    e_all, f_all, stress, per_atom = mod(*args, None, True)
    e_sel, f_sel, none_stress, per_sel = mod(*args, keep.to(dev), False)
    assert none_stress.shape[0] == 0 and per_sel.shape[0] == int(keep.sum())

    eps = torch.zeros(3, 3, dtype=torch.float64, requires_grad=True)
    r64 = pos.double().requires_grad_(True)
    strain = torch.eye(3, dtype=torch.float64) + eps
    atomic = opet.pet_atomic_energies(p64, hypers, r64 @ strain, (cell.double() @ strain)[None], i, j, s.long(), z,
                                      sysidx.long())[:, 0] * scale
    g_r, g_eps = torch.autograd.grad(atomic.sum(), [r64, eps], retain_graph=True)
    base = comp.double()[z.long()]
    vol = float(torch.det(cell.double()).abs())
    assert abs(float(e_all[0]) - float(atomic.sum() + base.sum())) / abs(float(atomic.sum() + base.sum())) < TOL
    assert relmax(-f_all.cpu().numpy(), g_r.numpy()) < TOL
    assert relmax(stress[0].cpu().numpy(), (g_eps / vol).numpy()) < TOL
    assert relmax(per_atom.cpu().numpy(), (atomic + base).detach().numpy()) < TOL
    (g_sel,) = torch.autograd.grad(atomic[keep].sum(), [r64])
    e_ref = float(atomic[keep].sum() + base[keep].sum())
    assert abs(float(e_sel[0]) - e_ref) / abs(e_ref) < TOL
    assert relmax(-f_sel.cpu().numpy(), g_sel.numpy()) < TOL
    assert relmax(per_sel.cpu().numpy(), (atomic + base)[keep].detach().numpy()) < TOL


# ---------------------------------------------------------------------------------------------------------
# the three calls are functions of their arguments
# ---------------------------------------------------------------------------------------------------------
def _heads_oracle(params, target, block, nf, ef_nef, mask, cf):
    """backend.py:651-777 in fp64 torch on given features: node + cutoff-weighted masked edge predictions."""
    silu = torch.nn.functional.silu
    lin = lambda x, k: torch.nn.functional.linear(x, params[k + ".weight"], params[k + ".bias"])  # noqa: E731
    hn = silu(lin(silu(lin(nf, f"node_heads.{target}.0.0")), f"node_heads.{target}.0.2"))
    he = silu(lin(silu(lin(ef_nef, f"edge_heads.{target}.0.0")), f"edge_heads.{target}.0.2"))
    pn = lin(hn, f"node_last_layers.{target}.0.{block}")
    pe = lin(he, f"edge_last_layers.{target}.0.{block}")
    pe = torch.where(mask[..., None], pe, torch.zeros_like(pe)) * cf[..., None]
    return pn + pe.sum(1), hn, he


def test_predict_is_a_function_of_the_features_it_is_given(dev):
    """Edit the features between calculate_features and predict (what a LoRA / finetune hook or a diagnostic does):
    predictions, their gradient w.r.t. the edited features and the returned last-layer features follow the edit."""
    be, hypers = _backend(dev)
    pos, z, cell = opet.random_box(60, 7)
    i, j, s, zz, sysidx = _inputs(pos.numpy(), z.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"], dev)
    cells = cell[None].to(dev)
    batch = be.preprocess(pos.to(dev), i, j, zz, cells, s, sysidx, 1.0)
    nodes, edges = be.calculate_features(batch)
    gen = torch.Generator().manual_seed(0)
    nf = (1.3 * nodes[0].cpu() + 0.2 * torch.randn(nodes[0].shape, generator=gen)).to(dev).requires_grad_(True)
    ef = (0.7 * edges[0].cpu() - 0.1 * torch.randn(edges[0].shape, generator=gen)).to(dev).requires_grad_(True)
    pred, node_ll, edge_ll = be.predict([nf], [ef], batch, cells, sysidx, ["energy"])
    w = torch.rand(60, 1, generator=gen).to(dev)
    g_nf, g_ef = torch.autograd.grad((pred["energy"][0] * w).sum(), [nf, ef])
    p64 = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    nf64 = nf.detach().cpu().double().requires_grad_(True)
    ef64 = ef.detach().cpu().double().requires_grad_(True)
    mask = batch["padding_mask"].cpu()
    ref, hn, he = _heads_oracle(p64, "energy", "energy", nf64, ef64, mask, batch["cutoff_factors"].cpu().double())
    r_nf, r_ef = torch.autograd.grad((ref * w.cpu().double()).sum(), [nf64, ef64])
    assert relmax(pred["energy"][0].detach().cpu().numpy(), ref.detach().numpy()) < TOL
    assert relmax(g_nf.cpu().numpy(), r_nf.numpy()) < TOL and relmax(g_ef.cpu().numpy(), r_ef.numpy()) < TOL
    assert float(g_ef.cpu()[~mask].abs().max()) == 0.0  # pads get no gradient
    assert relmax(node_ll["energy"][0].cpu().numpy(), hn.detach().numpy()) < TOL
    he_ref = torch.where(mask[..., None], he.detach(), torch.zeros_like(he.detach()))
    assert relmax(edge_ll["energy"][0].cpu().numpy(), he_ref.numpy()) < TOL
    # and the un-edited features still give the un-edited answer
    plain, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy"])
    assert relmax(plain["energy"][0].detach().cpu().numpy(), pred["energy"][0].detach().cpu().numpy()) > 1e-2


def test_calculate_features_accepts_the_reference_batch_data(dev, golden_dir):
    """``batch_data`` produced by the REFERENCE's preprocess (golden batch_box64.npz: shuffled non-strict list, pads
    that replicate edge 0, unique pad ids in reverse_neighbor_index) goes straight into calculate_features / predict:
    per-atom energies equal the reference's for the same box (pet_default_box64.npz)."""
    be, hypers = _backend(dev)
    b = dict(np.load(os.path.join(golden_dir, "batch_box64.npz")))
    g = dict(np.load(os.path.join(golden_dir, "pet_default_box64.npz")))
    batch = {k: torch.tensor(v).to(dev) for k, v in b.items() if not k.startswith("in_")}
    batch["edge_vectors"].requires_grad_(True)
    nodes, edges = be.calculate_features(batch)
    cells = torch.tensor(b["in_cells"]).float().to(dev)
    sysidx = torch.tensor(b["in_system_indices"]).to(dev)
    pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy"])
    assert relmax(pred["energy"][0].cpu().detach().numpy(), g["atomic_f64"]) < TOL
    assert relmax(nodes[0].cpu().detach().numpy(), g["node_features_f64"]) < TOL
    # dE/d(edge_vectors) through predict^T and features^T, scattered by hand = dE/dR of the golden (structures.py:220)
    (g_ev,) = torch.autograd.grad(pred["energy"][0].sum(), batch["edge_vectors"])
    assert float(g_ev[~batch["padding_mask"]].abs().max()) == 0.0


def test_several_targets_blocks_and_properties(dev):
    """backend.py:171-217, :689-777: a second target with two blocks of 3 and 6 properties next to the energy, every
    block against the fp64 oracle, gradients w.r.t. positions through all three calls; the non-conservative stress
    post-processing (backend.py:780-813)."""
    from metatrain_amd.pet import PETBackend, default_hypers

    hypers = default_hypers()
    targets = {"energy": 1, "multi": {"a": 3, "b": 6}, "non_conservative_stress": 9}
    be = PETBackend(hypers, [1, 6, 7, 8])
    be.add_output("energy", {"energy": [1]})
    be.add_output("multi", {"a": [3], "b": [3, 2]})
    be.add_output("non_conservative_stress", {"non_conservative_stress": [3, 3, 1]})
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], targets)
    be.load_state_dict(params, strict=True)
    be = be.to(dev).eval()
    pos, z, cell = opet.random_box(50, 9)
    i, j, s, zz, sysidx = _inputs(pos.numpy(), z.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"], dev)
    cells = cell[None].to(dev)
    p = pos.to(dev).requires_grad_(True)
    batch = be.preprocess(p, i, j, zz, cells, s, sysidx, 1.0)
    nodes, edges = be.calculate_features(batch)
    pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy", "multi", "non_conservative_stress"])
    assert [tuple(t.shape) for t in pred["multi"]] == [(50, 3), (50, 6)] and pred["energy"][0].shape == (50, 1)
    assert pred["non_conservative_stress"][0].shape == (50, 3, 3, 1)
    gen = torch.Generator().manual_seed(1)
    wa, wb = torch.randn(50, 3, generator=gen), torch.randn(50, 6, generator=gen)
    (gp,) = torch.autograd.grad((pred["multi"][0] * wa.to(dev)).sum() + (pred["multi"][1] * wb.to(dev)).sum()
                                + pred["energy"][0].sum(), p)
    p64 = opet.synthetic_params(hypers, [1, 6, 7, 8], targets, 0, torch.float64)
    r = pos.double().requires_grad_(True)
    args = (hypers, r, cell[None].double(), i.cpu(), j.cpu(), s.cpu().long(), z, sysidx.cpu())
    ra = opet.pet_atomic_energies(p64, *args, "multi", "a")
    rb = opet.pet_atomic_energies(p64, *args, "multi", "b")
    re = opet.pet_atomic_energies(p64, *args, "energy")
    rs = opet.pet_atomic_energies(p64, *args, "non_conservative_stress")
    (gr,) = torch.autograd.grad((ra * wa.double()).sum() + (rb * wb.double()).sum() + re.sum(), r)
    assert relmax(pred["multi"][0].detach().cpu().numpy(), ra.detach().numpy()) < TOL
    assert relmax(pred["multi"][1].detach().cpu().numpy(), rb.detach().numpy()) < TOL
    assert relmax(pred["energy"][0].detach().cpu().numpy(), re.detach().numpy()) < TOL
    assert relmax(gp.cpu().numpy(), gr.numpy()) < TOL
    t = rs.detach().reshape(-1, 3, 3, 1) / float(torch.det(cell.double()).abs())
    assert relmax(pred["non_conservative_stress"][0].detach().cpu().numpy(), ((t + t.transpose(1, 2)) / 2).numpy()) < TOL


def test_scripted_backend_equals_eager_and_survives_save_load(dev):
    """The mirror under torch.jit.script (what PET.export does to the module that owns it, pet/model.py:990-1021):
    same numbers as eager, forces by autograd through the three scripted calls, after a save / load round trip."""
    import io

    be, hypers = _backend(dev)
    pos, z, cell = opet.random_box(45, 4)
    i, j, s, zz, sysidx = _inputs(pos.numpy(), z.numpy(), cell.numpy(), [True] * 3, hypers["cutoff"], dev)
    cells = cell[None].to(dev)

    def run(module):
        p = pos.to(dev).requires_grad_(True)
        batch = module.preprocess(p, i, j, zz, cells, s, sysidx, 1.0)
        nodes, edges = module.calculate_features(batch)
        pred, _, _ = module.predict(nodes, edges, batch, cells, sysidx, ["energy"])
        (g,) = torch.autograd.grad([pred["energy"][0].sum()], [p])
        return pred["energy"][0].detach(), g

    e0, g0 = run(be)
    buf = io.BytesIO()
    torch.jit.save(torch.jit.script(be), buf)
    buf.seek(0)
    mod = torch.jit.load(buf, map_location=dev)
    e1, g1 = run(mod)
    assert torch.equal(e0, e1) and torch.equal(g0, g1)
    params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float64)
    r = pos.double().requires_grad_(True)
    ref = opet.pet_atomic_energies(params, hypers, r, cell[None].double(), i.cpu(), j.cpu(), s.cpu().long(), z, sysidx.cpu())
    (gr,) = torch.autograd.grad(ref.sum(), r)
    assert relmax(e1.cpu().numpy(), ref.detach().numpy()) < TOL and relmax(g1.cpu().numpy(), gr.numpy()) < TOL


def test_training_through_the_mirror_with_a_stress_term(golden_dir):
    """The reference's strain trick under create_graph (utils/evaluate_model.py:305-321: positions @ strain, cell @ strain,
    gradient w.r.t. strain) through the mirror in train() mode, a loss on energies, dE/dR and dE/dstrain,
    loss.backward() -> parameter.grad; against the fp64 oracle's double backward."""
    from metatrain_amd.pet import PETBackend, default_hypers

    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(golden_dir, "batch_two_systems.npz")))
    t = lambda k: torch.tensor(g[k])  # noqa: E731
    hypers, types = default_hypers(), [1, 6, 7, 8]
    params = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float32)
    sys_cpu = t("in_system_indices").long()
    n, n_sys = len(sys_cpu), int(sys_cpu.max()) + 1
    gen = torch.Generator().manual_seed(9)
    tgt_g, tgt_s = 0.3 * torch.randn(n, 3, generator=gen), torch.randn(n_sys, 3, 3, generator=gen)

    def loss_of(backend_like, dtype, device):
        pos0 = t("in_positions").to(device, dtype).requires_grad_(True)
        strain = torch.eye(3, dtype=dtype, device=device).repeat(n_sys, 1, 1).requires_grad_(True)
        sysidx = sys_cpu.to(device)
        pos = (pos0[:, None, :] @ strain[sysidx]).squeeze(1)
        cells = t("in_cells").to(device, dtype) @ strain
        atomic = backend_like(pos, cells, sysidx)
        g_pos, g_strain = torch.autograd.grad(atomic.sum(), [pos0, strain], create_graph=True)
        return (atomic.sum() ** 2 * 1e-3 + ((g_pos - tgt_g.to(device, dtype)) ** 2).sum()
                + ((g_strain - tgt_s.to(device, dtype)) ** 2).sum())

    be = PETBackend(hypers, types)
    be.add_output("energy", {"energy": [1]})
    be.load_state_dict(params, strict=True)
    be = be.to(dev)

    def hip(pos, cells, sysidx):
        batch = be.preprocess(pos, t("in_centers").to(dev), t("in_neighbors").to(dev), t("in_species").to(dev), cells,
                              t("in_cell_shifts").to(dev), sysidx, 1.0)
        nf, ef = be.calculate_features(batch)
        return be.predict(nf, ef, batch, cells, sysidx, ["energy"])[0]["energy"][0][:, 0]

    loss = loss_of(hip, torch.float32, dev)
    loss.backward()
    p64 = {k: (v if k == "species_to_species_index" else v.double().clone().requires_grad_(True)) for k, v in params.items()}

    def oracle(pos, cells, sysidx):
        return opet.pet_atomic_energies(p64, hypers, pos, cells, t("in_centers"), t("in_neighbors"), t("in_cell_shifts"),
                                        t("in_species"), sysidx, "energy")[:, 0]

    l_ref = loss_of(oracle, torch.float64, torch.device("cpu"))
    keys = [k for k in p64 if k != "species_to_species_index"]
    ref = dict(zip(keys, torch.autograd.grad(l_ref, [p64[k] for k in keys], allow_unused=True)))
    assert abs(float(loss) - float(l_ref)) / abs(float(l_ref)) < 1e-5
    named = dict(be.named_parameters())
    worst = 0.0
    for k in keys:
        if ref[k] is None:
            continue
        scale = float(ref[k].abs().max())
        if scale > 1e-12:
            worst = max(worst, float((named[k].grad.cpu().double() - ref[k]).abs().max()) / scale)
    assert worst < 1e-5, worst


def test_several_targets_against_the_reference_golden(dev, golden_dir):
    """The same three targets against what the imported REFERENCE produced (``make_golden.py --multitarget``,
    ``pet_multitarget_box50.npz``): every block's per-atom predictions, the processed non-conservative stress and dE/dR
    of the weighted sum, through the mirror's three calls."""
    from metatrain_amd.pet import PETBackend, default_hypers

    g = dict(np.load(os.path.join(golden_dir, "pet_multitarget_box50.npz")))
    hypers = default_hypers()
    targets = {"energy": 1, "multi": {"a": 3, "b": 6}, "non_conservative_stress": 9}
    be = PETBackend(hypers, [1, 6, 7, 8])
    be.add_output("energy", {"energy": [1]})
    be.add_output("multi", {"a": [3], "b": [3, 2]})
    be.add_output("non_conservative_stress", {"non_conservative_stress": [3, 3, 1]})
    be.load_state_dict(opet.synthetic_params(hypers, [1, 6, 7, 8], targets), strict=True)
    be = be.to(dev).eval()
    t = lambda k: torch.tensor(g[k]).to(dev)  # noqa: E731
    p = t("in_positions").float().requires_grad_(True)
    cells, sysidx = t("in_cells").float(), t("in_system_indices")
    batch = be.preprocess(p, t("in_centers"), t("in_neighbors"), t("in_species"), cells, t("in_cell_shifts"), sysidx, 1.0)
    nodes, edges = be.calculate_features(batch)
    pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy", "multi", "non_conservative_stress"])
    (gp,) = torch.autograd.grad((pred["multi"][0] * t("wa").float()).sum() + (pred["multi"][1] * t("wb").float()).sum()
                                + pred["energy"][0].sum(), p)
    assert relmax(pred["energy"][0].detach().cpu().numpy(), g["energy"]) < TOL
    assert relmax(pred["multi"][0].detach().cpu().numpy(), g["multi_a"]) < TOL
    assert relmax(pred["multi"][1].detach().cpu().numpy(), g["multi_b"]) < TOL
    assert relmax(pred["non_conservative_stress"][0].detach().cpu().numpy(), g["non_conservative_stress"]) < TOL
    assert relmax(gp.cpu().numpy(), g["grad"]) < TOL

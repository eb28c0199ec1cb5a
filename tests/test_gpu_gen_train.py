"""GPU tests of the size-generic TRAINING pass (csrc/gen_train.hip; VERDICT r2 missing #1 / #3): parameter gradients of the
energy term and of the force-loss term (the reference's double backward, ``utils/output_gradient.py:34-40`` +
``pet/trainer.py:462``) for models the tuned second-order pass does not serve -- other model sizes incl. ``d_node == d_pet``
and the reference's minimal hypers (``pet/tests/test_basic.py:22-32``), PostLN transformer layers
(``transformer.py:236-262``) and the residual featuriser (``backend.py:589-649``) at the DEFAULT size, and the
combination old checkpoints upgrade to (``pet/checkpoints.py:190-205``) -- against torch's autograd through the fp64
oracle, through the C ABI (``pet_backward_train`` / ``pet_backward_train2``) and the native ``TrainStep``."""
import numpy as np
import pytest
import torch

from oracle import pet as opet
from test_gpu_train import _inputs, _oracle_param_grads, _oracle_second_order

pytestmark = pytest.mark.gpu
TOL = 1e-5
TYPES = [1, 6, 7, 8]
CASES = {
    "s64": dict(d_pet=64, d_node=128, d_feedforward=128, d_head=64, num_heads=4),
    "flat32": dict(d_pet=32, d_node=32, d_feedforward=48, d_head=24, num_heads=2),
    "flat32_legacy": dict(d_pet=32, d_node=32, d_feedforward=48, d_head=24, num_heads=2, normalization="LayerNorm",
                          activation="SiLU", transformer_type="PostLN", featurizer_type="residual"),
    "minimal": dict(d_pet=1, d_node=1, d_feedforward=1, d_head=1, num_heads=1, num_attention_layers=1, num_gnn_layers=1),
    "default_postln": dict(transformer_type="PostLN"),
    "default_residual": dict(featurizer_type="residual"),
    "default_legacy": dict(normalization="LayerNorm", activation="SiLU", transformer_type="PostLN", featurizer_type="residual"),
    "s64_layernorm": dict(d_pet=64, d_node=128, d_feedforward=128, d_head=64, num_heads=4, normalization="LayerNorm"),
    # adaptive cutoff ('solver', structures.py:225-263): the pair cutoffs move with the positions, so the force-loss term
    # needs the implicit-function tangent of the atomic cutoffs (so.hip geometry_tangent) on the generic pass too
    "s64_adaptive": dict(d_pet=64, d_node=128, d_feedforward=128, d_head=64, num_heads=4, num_neighbors_adaptive=6.0),
    "default_residual_adaptive": dict(featurizer_type="residual", num_neighbors_adaptive=6.0),
    "s64_adaptive_grid": dict(d_pet=64, d_node=128, d_feedforward=128, d_head=64, num_heads=4, num_neighbors_adaptive=6.0,
                              adaptive_cutoff_method="grid"),
}


def _setup(golden_dir, tag, case="batch_two_systems.npz"):
    from metatrain_amd import runtime as rt

    dev = torch.device("cuda:0")
    hypers = dict(opet.DEFAULT_HYPERS, **CASES[tag])
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    inp = _inputs(golden_dir, case)
    model = rt.HipModel(hypers, TYPES)
    model.load({k: v.to(dev) for k, v in params.items()}, "energy")
    graph = rt.HipGraph(model, inp["positions"].float().to(dev), inp["cells"].float().to(dev),
                        inp["centers"].to(dev), inp["neighbors"].to(dev), inp["cell_shifts"].to(dev),
                        inp["species"].to(dev), inp["system_indices"].int().to(dev))
    return rt, dev, hypers, params, inp, model, graph


# d_pet = 1 is a degenerate model numerically: RMSNorm of ONE feature is sign(x) (x / sqrt(x^2 + 1.2e-7)), every gradient
# is a single sum with heavy cancellation. The reference's own arithmetic (torch, fp32, same weights and inputs) misses its
# fp64 values by 2e-4 .. 9e-3 on these gradients (edge_embedder.bias 9.2e-3, compress.0.bias 9.1e-3, norm / mlp weights
# 4e-4: measured with the oracle evaluated in fp32); the HIP pass sits at 1e-5 (energy term) / 8e-4 (force-loss term).
LOOSE = {"minimal": 2e-3}
# the grid adaptive cutoff carries the tangent through a soft-max over probe cutoffs: its force-loss gradients sit at 1.3e-5
LOOSE_FORCE = dict(LOOSE, s64_adaptive_grid=2 * TOL)


def _compare(got, ref, model, what, tol=TOL):
    worst = {}
    for k, r in ref.items():
        r = r.numpy()
        g = got[k].cpu().numpy().astype(np.float64)
        if g.shape != r.shape:   # activation = "SiLU": the model holds [W; W]; d/dW = the sum of the halves' gradients
            assert g.shape[0] == 2 * r.shape[0], k
            g = g[: r.shape[0]] + g[r.shape[0]:]
        scale = np.abs(r).max()
        err = np.abs(g - r).max()
        worst[k] = err / scale if scale > 1e-12 else err
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print(what, [(f"{v:.2e}", k) for k, v in top])
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, f"{what}: parameter gradients off: {bad}"
    return max(worst.values())


@pytest.mark.parametrize("tag", list(CASES))
def test_energy_term_parameter_gradients(golden_dir, tag):
    rt, dev, hypers, params, inp, model, graph = _setup(golden_dir, tag)
    n = inp["positions"].shape[0]
    seed_w = torch.rand(n, generator=torch.Generator().manual_seed(7)) + 0.5
    ref = _oracle_param_grads(params, hypers, inp, seed_w)
    fw = rt.HipForward(model, graph, train=True)
    model.zero_grad()
    fw.forward()
    gpos = fw.backward_train(seed_w.to(dev), want_position_grad=True)
    _compare(model.grads(), ref, model, f"{tag} energy term", LOOSE.get(tag, TOL))
    inf = rt.HipForward(model, graph)
    a = inf.forward() if hypers["featurizer_type"] != "residual" or model.hypers["d_pet"] != 128 else None
    if a is not None:   # (the compiled size serves the residual featuriser's inference through the staged calls only)
        np.testing.assert_allclose(gpos.cpu().numpy(), inf.backward(seed_w.to(dev)).cpu().numpy(), rtol=0,
                                   atol=2e-6 * float(gpos.abs().max()))


@pytest.mark.parametrize("tag", list(CASES))
def test_force_loss_parameter_gradients(golden_dir, tag):
    rt, dev, hypers, params, inp, model, graph = _setup(golden_dir, tag)
    n = inp["positions"].shape[0]
    gen = torch.Generator().manual_seed(11)
    nu = torch.rand(n, generator=gen) - 0.5
    u = torch.randn(n, 3, generator=gen)
    ref, tan_ref, g_ref = _oracle_second_order(params, hypers, inp, nu, u)
    fw = rt.HipForward(model, graph, train=True)
    model.zero_grad()
    atomic = fw.forward()
    ones = torch.ones(n, device=dev)
    gpos = fw.backward(ones)
    assert np.abs(gpos.cpu().numpy() - g_ref.numpy()).max() < TOL * np.abs(g_ref.numpy()).max()
    tan = fw.backward_train2(ones, nu.to(dev), u.to(dev), want_tangent=True)
    assert np.abs(tan.cpu().numpy() - tan_ref.numpy()).max() / np.abs(tan_ref.numpy()).max() < TOL
    lhs, rhs = float(tan.double().sum()), float((u.to(dev).double() * gpos.double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(rhs))
    _compare(model.grads(), ref, model, f"{tag} force-loss term", LOOSE_FORCE.get(tag, TOL))
    # bit-reproducible: fixed summation orders, no atomics
    first = model.flat_grad().clone()
    model.zero_grad()
    fw.forward()
    fw.backward(ones)
    fw.backward_train2(ones, nu.to(dev), u.to(dev))
    assert torch.equal(model.flat_grad(), first)


@pytest.mark.parametrize("tag", ["s64", "default_legacy"])
def test_native_training_steps_reduce_the_loss_and_follow_torch(golden_dir, tag):
    """``TrainStep`` (zero_grad, forward, dE/dR, MSE(E/atom) + MSE(dE/dR), second-order pass, clip, Adam) for three steps
    against the same steps of torch.optim.Adam on the fp64 oracle."""
    from metatrain_amd.pet.trainer import TrainStep

    rt, dev, hypers, params, inp, model, graph = _setup(golden_dir, tag)
    s = inp["system_indices"].long()
    n_sys = int(s.max()) + 1
    n_atoms = torch.bincount(s, minlength=n_sys).float()
    gen = torch.Generator().manual_seed(3)
    te = torch.randn(n_sys, generator=gen) * n_atoms
    tg = torch.randn(len(s), 3, generator=gen) * 0.3
    lr = 1e-4
    p64 = {k: (v if k == "species_to_species_index" else v.double().clone().requires_grad_(True)) for k, v in params.items()}
    leaves = [v for k, v in p64.items() if k != "species_to_species_index"]
    opt = torch.optim.Adam(leaves, lr=lr)
    ref_losses = []
    for _ in range(3):
        opt.zero_grad()
        pos = inp["positions"].double().clone().requires_grad_(True)
        atomic = opet.pet_atomic_energies(p64, hypers, pos, inp["cells"].double(), inp["centers"], inp["neighbors"],
                                          inp["cell_shifts"], inp["species"], s, "energy")[:, 0]
        e = torch.zeros(n_sys, dtype=torch.float64).index_add(0, s, atomic)
        (g,) = torch.autograd.grad(e.sum(), pos, create_graph=True)
        loss = (((e - te.double()) / n_atoms.double()) ** 2).mean() + ((g - tg.double()) ** 2).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(leaves, 1.0)
        opt.step()
        ref_losses.append(float(loss.detach()))
    fw = rt.HipForward(model, graph, train=True)
    step = TrainStep(model, {"learning_rate": lr, "warmup_fraction": 0.0, "num_epochs": 10**9})
    losses = [float(step(graph, fw, te.to(dev), n_atoms.to(dev), tg.to(dev))["loss"]) for _ in range(3)]
    np.testing.assert_allclose(losses, ref_losses, rtol=5e-5)
    assert (losses[-1] < losses[0]) == (ref_losses[-1] < ref_losses[0])


@pytest.mark.parametrize("tag,extra", [
    ("cond8_flat", dict(d_pet=8, d_head=8, d_node=8, d_feedforward=8, num_heads=1, num_attention_layers=1, num_gnn_layers=1)),
    ("cond8_residual", dict(d_pet=8, d_head=8, d_node=8, d_feedforward=8, num_heads=1, num_attention_layers=1, num_gnn_layers=1,
                            featurizer_type="residual")),
    ("cond8_16", dict(d_pet=8, d_head=8, d_node=16, d_feedforward=8, num_heads=1, num_attention_layers=1, num_gnn_layers=1)),
    ("cond_default_residual", dict(featurizer_type="residual")),
])
def test_conditioned_models_train_on_the_size_generic_path(golden_dir, tag, extra):
    """``pet/tests/test_conditioning.py:28-40, 106-147`` trains conditioned models of d_pet = 8 (d_node 8 and 16, both
    featurisers) before looking at their outputs: every parameter gradient, the six conditioning tensors included, of the
    energy-only pass and of the force-loss pass against torch's (double) backward through the fp64 oracle."""
    from metatrain_amd import runtime as rt
    from test_gpu_train import _oracle_param_grads_cond

    dev = torch.device("cuda:0")
    hypers = dict(opet.DEFAULT_HYPERS, system_conditioning=True, **extra)
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    inp = _inputs(golden_dir, "batch_two_systems.npz")
    inp["charge"], inp["spin_multiplicity"] = torch.tensor([-2, 3]), torch.tensor([1, 4])
    n = inp["positions"].shape[0]
    gen = torch.Generator().manual_seed(17)
    nu = torch.rand(n, generator=gen) - 0.5
    u = torch.randn(n, 3, generator=gen)
    model = rt.HipModel(hypers, TYPES)
    model.load({k: v.to(dev) for k, v in params.items()}, "energy")
    graph = rt.HipGraph(model, inp["positions"].float().to(dev), inp["cells"].float().to(dev),
                        inp["centers"].to(dev), inp["neighbors"].to(dev), inp["cell_shifts"].to(dev),
                        inp["species"].to(dev), inp["system_indices"].int().to(dev))
    graph.set_conditioning(inp["charge"].to(dev), inp["spin_multiplicity"].to(dev), inp["system_indices"].to(dev))
    fw = rt.HipForward(model, graph, train=True)
    cond_keys = [k for k in params if k.startswith("system_conditioning.")]
    assert len(cond_keys) == 6
    w = torch.rand(n, generator=gen) + 0.5
    ref1 = _oracle_param_grads_cond(params, hypers, inp, w)
    model.zero_grad()
    fw.forward()
    fw.backward_train(w.to(dev))
    got = model.grads()
    _compare(got, ref1, model, f"{tag} energy term")
    assert all(float(got[k].abs().max()) > 0 for k in cond_keys)
    ref2, tan_ref, g_ref = _oracle_second_order(params, hypers, inp, nu, u)
    model.zero_grad()
    fw.forward()
    ones = torch.ones(n, device=dev)
    gpos = fw.backward(ones)
    assert np.abs(gpos.cpu().numpy() - g_ref.numpy()).max() < TOL * np.abs(g_ref.numpy()).max()
    tan = fw.backward_train2(ones, nu.to(dev), u.to(dev), want_tangent=True)
    assert np.abs(tan.cpu().numpy() - tan_ref.numpy()).max() / np.abs(tan_ref.numpy()).max() < TOL
    _compare(model.grads(), ref2, model, f"{tag} force-loss term")


def test_default_size_trains_on_a_graph_with_more_than_127_neighbours():
    """structures.py:292-294 pads to any max(num_neighbors); the tuned second-order pass serves 127. A graph with a denser
    atom runs its training forward, both reverse passes and the weight gradients on the size-generic path: parameter
    gradients of the energy term and of the force-loss term against the oracle's (double) backward."""
    from metatrain_amd import runtime as rt
    from oracle import nl as onl

    dev = torch.device("cuda:0")
    hypers = dict(opet.DEFAULT_HYPERS)
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    gen = torch.Generator().manual_seed(4)
    n = 150
    pos = torch.rand(n, 3, generator=gen, dtype=torch.float64) * 2.4   # every pair within the 4.5 A cutoff
    cell = torch.eye(3, dtype=torch.float64) * 30.0
    z = torch.tensor(TYPES)[torch.randint(0, 4, (n,), generator=gen)]
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [False] * 3, hypers["cutoff"])
    assert np.bincount(i, minlength=n).max() > 127
    inp = {"positions": pos, "cells": cell[None], "centers": torch.tensor(i).long(), "neighbors": torch.tensor(j).long(),
           "cell_shifts": torch.tensor(s).long(), "species": z, "system_indices": torch.zeros(n, dtype=torch.long)}
    nu = torch.rand(n, generator=gen) - 0.5
    u = torch.randn(n, 3, generator=gen)
    w = torch.rand(n, generator=gen) + 0.5
    model = rt.HipModel(hypers, TYPES)
    model.load({k: v.to(dev) for k, v in params.items()}, "energy")
    graph = rt.HipGraph(model, pos.float().to(dev), inp["cells"].float().to(dev), inp["centers"].to(dev),
                        inp["neighbors"].to(dev), inp["cell_shifts"].to(dev), z.to(dev),
                        inp["system_indices"].int().to(dev))
    fw = rt.HipForward(model, graph, train=True)
    ref1 = _oracle_param_grads(params, hypers, inp, w)
    model.zero_grad()
    fw.forward()
    fw.backward_train(w.to(dev))
    _compare(model.grads(), ref1, model, "dense graph, energy term")
    ref2, tan_ref, g_ref = _oracle_second_order(params, hypers, inp, nu, u)
    model.zero_grad()
    fw.forward()
    ones = torch.ones(n, device=dev)
    gpos = fw.backward(ones)
    assert np.abs(gpos.cpu().numpy() - g_ref.numpy()).max() < TOL * np.abs(g_ref.numpy()).max()
    tan = fw.backward_train2(ones, nu.to(dev), u.to(dev), want_tangent=True)
    assert np.abs(tan.cpu().numpy() - tan_ref.numpy()).max() / np.abs(tan_ref.numpy()).max() < TOL
    _compare(model.grads(), ref2, model, "dense graph, force-loss term")


@pytest.mark.parametrize("tag", ["default", "s64", "default_legacy"])
def test_training_on_a_batch_of_isolated_atoms(tag):
    """A batch WITHOUT ANY EDGE (the isolated-atom reference structures of a dataset falling into one batch): the node path
    alone carries the gradients -- embeddings, centre tokens attending to themselves, centre MLPs, node heads -- every edge
    parameter gets exactly zero, dE/dR and the tangent energies are zero. Against the oracle's backward."""
    from metatrain_amd import runtime as rt

    dev = torch.device("cuda:0")
    hypers = dict(opet.DEFAULT_HYPERS, **({} if tag == "default" else CASES[tag]))
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    n = 5
    pos = torch.tensor([[0.0, 0, 0], [40.0, 0, 0], [0, 40.0, 0], [0, 0, 40.0], [40.0, 40.0, 0]])
    z = torch.tensor([1, 6, 7, 8, 6])
    sysidx = torch.tensor([0, 0, 1, 2, 2])
    e0 = torch.zeros(0, dtype=torch.long)
    inp = {"positions": pos, "cells": torch.zeros(3, 3, 3), "centers": e0, "neighbors": e0,
           "cell_shifts": torch.zeros((0, 3), dtype=torch.long), "species": z, "system_indices": sysidx}
    model = rt.HipModel(hypers, TYPES)
    model.load({k: v.to(dev) for k, v in params.items()}, "energy")
    graph = rt.HipGraph(model, pos.to(dev), inp["cells"].to(dev), e0.to(dev), e0.to(dev), inp["cell_shifts"].to(dev), z.to(dev),
                        sysidx.int().to(dev))
    assert graph.n_edges == 0
    w = torch.tensor([0.7, -1.3, 0.4, 2.0, 1.1])
    ref = _oracle_param_grads(params, hypers, inp, w)
    fw = rt.HipForward(model, graph, train=True)
    model.zero_grad()
    atomic = fw.forward()
    gpos = fw.backward_train(w.to(dev), want_position_grad=True)
    assert float(gpos.abs().max()) == 0.0
    got = model.grads()
    _compare(got, ref, model, f"{tag} isolated atoms, energy term")
    edge_keys = [k for k in ref if float(ref[k].abs().max()) == 0.0]
    assert edge_keys and all(float(got[k].abs().max()) == 0.0 for k in edge_keys)
    # the force-loss pass: no geometry to differentiate, the nu term alone
    model.zero_grad()
    fw.forward()
    ones = torch.ones(n, device=dev)
    assert float(fw.backward(ones).abs().max()) == 0.0
    tan = fw.backward_train2(ones, w.to(dev), torch.randn(n, 3, generator=torch.Generator().manual_seed(1)).to(dev),
                             want_tangent=True)
    assert float(tan.abs().max()) == 0.0
    _compare(model.grads(), ref, model, f"{tag} isolated atoms, force-loss pass")
    a_ref = opet.pet_atomic_energies({k: (v if k == "species_to_species_index" else v.double()) for k, v in params.items()},
                                     hypers, pos.double(), inp["cells"].double(), e0, e0, inp["cell_shifts"], z, sysidx,
                                     "energy")[:, 0]
    assert float((atomic.cpu().double() - a_ref).abs().max() / a_ref.abs().max()) < TOL


def test_mirror_trains_on_isolated_atoms():
    """The same through the mirror in train() mode: ``loss.backward()`` with an energy and a force term on an edge-free batch."""
    from metatrain_amd.pet import PETBackend

    dev = torch.device("cuda:0")
    hypers = dict(opet.DEFAULT_HYPERS)
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    be = PETBackend(hypers, TYPES)
    be.add_output("energy", {"energy": [1]})
    be.load_state_dict(params, strict=True)
    be = be.to(dev).train()
    pos = torch.tensor([[0.0, 0, 0], [40.0, 0, 0], [0, 40.0, 0]], device=dev, requires_grad=True)
    z = torch.tensor([1, 8, 6], device=dev)
    sysidx = torch.tensor([0, 0, 1], device=dev)
    e0 = torch.zeros(0, dtype=torch.long, device=dev)
    cells = torch.zeros(2, 3, 3, device=dev)
    batch = be.preprocess(pos, e0, e0, z, cells, torch.zeros((0, 3), dtype=torch.long, device=dev), sysidx, 1.0)
    nodes, edges = be.calculate_features(batch)
    pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy"])
    atomic = pred["energy"][0][:, 0]
    (grad,) = torch.autograd.grad(atomic.sum(), pos, create_graph=True)
    w = torch.tensor([0.5, -1.0, 2.0], device=dev)
    loss = (w * atomic).sum() + ((grad - 0.1) ** 2).sum()
    loss.backward()
    inp = {"positions": pos.detach().cpu(), "cells": cells.cpu(), "centers": e0.cpu(), "neighbors": e0.cpu(),
           "cell_shifts": torch.zeros((0, 3), dtype=torch.long), "species": z.cpu(), "system_indices": sysidx.cpu()}
    ref = _oracle_param_grads(params, hypers, inp, w.cpu())
    named = dict(be.named_parameters())
    got = {k: (named[k].grad if named[k].grad is not None else torch.zeros_like(named[k])) for k in ref}
    _compare(got, ref, None, "mirror, isolated atoms")
    assert float(grad.abs().max()) == 0.0

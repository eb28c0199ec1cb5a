"""
Generate the golden fixtures in this directory by IMPORTING the reference
(`/root/reference`, read-only) in the build container. The reference never
travels to the GPU box; only the ``*.npz`` files written here do.

Run from the repo root:   python tests/golden/make_golden.py

What is imported: the seven PET module files
``/root/reference/src/metatrain/pet/modules/{utilities,nef,adaptive_cutoff,
conditioning,transformer,structures,backend}.py`` are loaded UNCHANGED by path,
after three stub modules provide the type-annotation-only names they import
(``metatensor.torch.Labels``, ``metatomic.torch.{NeighborListOptions,System}``,
``metatrain.pet.documentation.ModelHypers``). Recipe: SURVEY.md Appendix A.

Fixtures written (all small):

``qm9_first5.npz``         inputs of the first five frames of the reference's
                           ``tests/resources/qm9_reduced_100.xyz`` + the five
                           regression energies hard-coded in
                           ``pet/tests/test_regression.py:66-74`` + the energies
                           this import produced with ``torch.manual_seed(0)``.
``batch_<case>.npz``       the 12 ``batch_data`` tensors of ``PETBackend.preprocess``
                           for: ``co2cell`` (2-atom 3.5 A cubic C/O cell of
                           ``pet/tests/test_backend.py:69-81``), ``box64``
                           (64-atom random box, shuffled non-strict NL with rc+1),
                           ``two_systems`` (two boxes with different cells).
``pet_tiny_<dtype>.npz``   E, dE/dR, per-atom E for reduced hypers (d_pet=16, ...)
                           with the synthetic weight generator, fp32 and fp64.
``pet_default_box64.npz``  E, dE/dR, per-atom E, node/edge features for default
                           hypers with the synthetic weight generator (seed 0).
``pet_default_box1000.npz`` E, dE/dR for a 1000-atom box,
                           fp32 and fp64 reference values.
"""

import importlib.util
import os
import random
import sys
import types

import numpy as np
import torch


HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/metatrain"


def import_reference_backend():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class _Placeholder:  # annotation-only names
        pass

    stub("metatensor")
    stub("metatensor.torch", Labels=_Placeholder)
    stub("metatomic")
    stub("metatomic.torch", NeighborListOptions=_Placeholder, System=_Placeholder)
    for pkg in ("metatrain", "metatrain.pet", "metatrain.pet.modules"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    stub("metatrain.pet.documentation", ModelHypers=dict)
    for f in (
        "utilities", "nef", "adaptive_cutoff", "conditioning", "transformer",
        "structures", "backend",
    ):
        spec = importlib.util.spec_from_file_location(
            f"metatrain.pet.modules.{f}", f"{REF}/pet/modules/{f}.py"
        )
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
    from metatrain.pet.modules.backend import PETBackend

    return PETBackend


def read_xyz_frames(path, n_frames):
    """Minimal extended-xyz reader for species + positions (non-periodic QM9)."""
    sym2z = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9}
    frames = []
    with open(path) as fh:
        lines = fh.readlines()
    k = 0
    while len(frames) < n_frames:
        n = int(lines[k])
        body = lines[k + 2 : k + 2 + n]
        z = [sym2z[ln.split()[0]] for ln in body]
        xyz = [[float(x) for x in ln.split()[1:4]] for ln in body]
        frames.append((np.array(z), np.array(xyz)))
        k += 2 + n
    return frames


def run_reference(backend, name, pos, cells, centers, neighbors, shifts, species, sysidx,
                  want_features=False):
    pos = pos.detach().clone().requires_grad_(True)
    batch = backend.preprocess(pos, centers, neighbors, species, cells, shifts, sysidx, 1.0)
    nf, ef = backend.calculate_features(batch)
    pred, _, _ = backend.predict(nf, ef, batch, cells, sysidx, [name])
    atomic = pred[name][0]
    n_sys = cells.shape[0]
    energies = torch.zeros(n_sys, atomic.shape[1], dtype=atomic.dtype).index_add(
        0, sysidx, atomic
    )
    (grad,) = torch.autograd.grad(energies.sum(), pos)
    out = {
        "energies": energies.detach().numpy(),
        "atomic": atomic.detach().numpy(),
        "grad": grad.numpy(),
    }
    if want_features:
        out["node_features"] = nf[0].detach().numpy()
        out["edge_features"] = ef[0].detach().numpy()
    return out, batch


def main_adaptive(method="solver"):
    """Adaptive-cutoff fixtures (``num_neighbors_adaptive``, "solver" method, or with
    ``--adaptive-grid`` the legacy "grid" method, files ``*_adaptive_grid_<case>.npz``):
    ``batch_adaptive_<case>.npz`` (12 ``batch_data`` tensors) and ``pet_adaptive_<case>.npz``
    (E, per-atom E, dE/dR in fp32 / fp64) from the reference modules."""
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    hypers = dict(opet.DEFAULT_HYPERS, num_neighbors_adaptive=12, adaptive_cutoff_method=method,
                  cutoff_width_adaptive=1.0)
    name = "adaptive" if method == "solver" else f"adaptive_{method}"
    p64, z64, c64 = opet.random_box(64, seed=1)
    p40, z40, c40 = opet.random_box(40, seed=2)
    tri = c40.clone(); tri[1, 0] = 2.0; tri[2, 1] = -1.5
    cases = {"box64": [(p64, z64, c64)], "two_systems": [(p64, z64, c64), (p40 + 30.0, z40, tri)]}
    for tag, systems in cases.items():
        pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l = [], [], [], [], [], [], []
        off = 0
        for k, (pos, z, cell) in enumerate(systems):
            i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), [True] * 3, hypers["cutoff"])
            pos_l.append(pos); z_l.append(z); cell_l.append(cell)
            i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off)
            s_l.append(torch.tensor(s)); sys_l.append(torch.full((len(z),), k, dtype=torch.long))
            off += len(z)
        pos = torch.cat(pos_l); z = torch.cat(z_l); cells = torch.stack(cell_l)
        i = torch.cat(i_l); j = torch.cat(j_l); s = torch.cat(s_l).long(); sysidx = torch.cat(sys_l)
        perm = torch.randperm(len(i), generator=torch.Generator().manual_seed(5))
        i, j, s = i[perm], j[perm], s[perm]
        inputs = {"in_" + k: v.numpy() for k, v in dict(
            positions=pos.double(), species=z, cells=cells.double(), centers=i.int(), neighbors=j.int(),
            cell_shifts=s.int(), system_indices=sysidx).items()}
        be = PETBackend(hypers, [1, 6, 7, 8])
        batch = be.preprocess(pos, i, j, z, cells, s, sysidx, hypers["cutoff_width_adaptive"])
        out = dict(inputs)
        out.update({k: v.numpy() for k, v in batch.items()})
        np.savez(os.path.join(HERE, f"batch_{name}_{tag}.npz"), **out)
        print(tag, "E0 =", len(i), "kept", len(batch["centers"]), "M =", batch["padding_mask"].shape[1],
              "cutoffs", batch["atomic_cutoffs_stats"].min().item(), batch["atomic_cutoffs_stats"].max().item())
        store = dict(inputs)
        for dtype in (torch.float32, torch.float64):
            params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, dtype)
            be = PETBackend(hypers, [1, 6, 7, 8])
            be.add_output("energy", {"energy": [1]})
            be = be.to(dtype).eval()
            be.load_state_dict(params, strict=True)
            p = pos.to(dtype).clone().requires_grad_(True)
            b = be.preprocess(p, i, j, z, cells.to(dtype), s, sysidx, hypers["cutoff_width_adaptive"])
            nf, ef = be.calculate_features(b)
            pred, _, _ = be.predict(nf, ef, b, cells.to(dtype), sysidx, ["energy"])
            atomic = pred["energy"][0]
            energies = torch.zeros(cells.shape[0], 1, dtype=dtype).index_add(0, sysidx, atomic)
            (grad,) = torch.autograd.grad(energies.sum(), p)
            sfx = {torch.float32: "f32", torch.float64: "f64"}[dtype]
            store[f"energies_{sfx}"] = energies.detach().numpy()
            store[f"atomic_{sfx}"] = atomic.detach().numpy()
            store[f"grad_{sfx}"] = grad.numpy()
            store[f"atomic_cutoffs_{sfx}"] = b["atomic_cutoffs_stats"].numpy()
            print(tag, sfx, "E =", energies.detach().numpy().ravel(), "|grad|max =", grad.abs().max().item())
        np.savez_compressed(os.path.join(HERE, f"pet_{name}_{tag}.npz"), **store)


def main():
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    hypers = dict(opet.DEFAULT_HYPERS)

    # ------------------------------------------------------------------ regression pin
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    backend = PETBackend(hypers, [1, 6, 7, 8])
    backend.add_output("mtt::U0", {"mtt::U0": [1]})
    backend.eval()
    frames = read_xyz_frames("/root/reference/tests/resources/qm9_reduced_100.xyz", 5)
    expected = np.array(  # pet/tests/test_regression.py:66-74
        [1.146098375320, 0.171331465244, 0.539504408836, 0.861489117146, 0.177449733019]
    )
    got = []
    store = {"expected_reference_test": expected}
    for k, (z, xyz) in enumerate(frames):
        i, j, s, _ = onl.neighbor_list(xyz, np.zeros((3, 3)), [False] * 3, hypers["cutoff"])
        pos = torch.tensor(xyz, dtype=torch.float32)
        res, _ = run_reference(
            backend, "mtt::U0", pos, torch.zeros(1, 3, 3),
            torch.tensor(i), torch.tensor(j), torch.tensor(s).long(),
            torch.tensor(z), torch.zeros(len(z), dtype=torch.long),
        )
        got.append(float(res["energies"][0, 0]))
        store[f"z{k}"] = z.astype(np.int32)
        store[f"pos{k}"] = xyz
    got = np.array(got)
    print("regression energies: got", got, "\n expected", expected)
    assert np.allclose(got, expected, rtol=1.3e-6, atol=1e-5), "reference import broken"
    store["reference_import_seed0"] = got
    np.savez(os.path.join(HERE, "qm9_first5.npz"), **store)

    # ------------------------------------------------------------------ batch_data
    def batch_case(tag, systems, cutoff_nl, shuffle_seed=None):
        pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l = [], [], [], [], [], [], []
        off = 0
        for k, (pos, z, cell, pbc) in enumerate(systems):
            i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), pbc, cutoff_nl)
            pos_l.append(pos); z_l.append(z); cell_l.append(cell)
            i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off)
            s_l.append(torch.tensor(s)); sys_l.append(torch.full((len(z),), k, dtype=torch.long))
            off += len(z)
        pos = torch.cat(pos_l); z = torch.cat(z_l); cells = torch.stack(cell_l)
        i = torch.cat(i_l); j = torch.cat(j_l); s = torch.cat(s_l); sysidx = torch.cat(sys_l)
        if shuffle_seed is not None:
            perm = torch.randperm(len(i), generator=torch.Generator().manual_seed(shuffle_seed))
            i, j, s = i[perm], j[perm], s[perm]
        be = PETBackend(hypers, [1, 6, 7, 8])
        batch = be.preprocess(pos, i, j, z, cells, s, sysidx, 1.0)
        out = {"in_" + k: v.numpy() for k, v in dict(
            positions=pos, species=z, cells=cells, centers=i, neighbors=j, cell_shifts=s,
            system_indices=sysidx).items()}
        out.update({k: v.numpy() for k, v in batch.items()})
        np.savez(os.path.join(HERE, f"batch_{tag}.npz"), **out)
        print(tag, "E =", len(i), "kept", len(batch["centers"]), "M =", batch["padding_mask"].shape[1])

    pbc = [True] * 3
    co = (torch.tensor([[0.0, 0.0, 0.0], [1.5, 1.5, 1.5]]), torch.tensor([6, 8], dtype=torch.int32),
          3.5 * torch.eye(3), pbc)
    batch_case("co2cell", [co], hypers["cutoff"])
    p64, z64, c64 = opet.random_box(64, seed=1)
    batch_case("box64", [(p64, z64, c64, pbc)], hypers["cutoff"] + 1.0, shuffle_seed=3)
    p40, z40, c40 = opet.random_box(40, seed=2)
    tri = c40.clone(); tri[1, 0] = 2.0; tri[2, 1] = -1.5
    batch_case("two_systems", [(p64, z64, c64, pbc), (p40 + 30.0, z40, tri, pbc)],
               hypers["cutoff"], shuffle_seed=4)

    # ------------------------------------------------------------------ E / dE/dR
    def pet_case(tag, hyp, systems, dtypes, seed=0, want_features=False, save_params=False):
        store = {}
        for dtype in dtypes:
            params = opet.synthetic_params(hyp, [1, 6, 7, 8], {"energy": 1}, seed, dtype)
            be = PETBackend(hyp, [1, 6, 7, 8])
            be.add_output("energy", {"energy": [1]})
            be = be.to(dtype).eval()
            missing = be.load_state_dict(params, strict=True)
            assert not missing.missing_keys and not missing.unexpected_keys
            assert list(be.state_dict().keys()) == list(params.keys()), "schema order"
            pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l = [], [], [], [], [], [], []
            off = 0
            for k, (pos, z, cell) in enumerate(systems):
                i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(),
                                               [True] * 3, hyp["cutoff"])
                pos_l.append(pos.to(dtype)); z_l.append(z); cell_l.append(cell.to(dtype))
                i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off)
                s_l.append(torch.tensor(s)); sys_l.append(torch.full((len(z),), k, dtype=torch.long))
                off += len(z)
            args = (torch.cat(pos_l), torch.stack(cell_l), torch.cat(i_l), torch.cat(j_l),
                    torch.cat(s_l).long(), torch.cat(z_l), torch.cat(sys_l))
            res, _ = run_reference(be, "energy", *args, want_features=want_features)
            sfx = {torch.float32: "f32", torch.float64: "f64"}[dtype]
            for k, v in res.items():
                store[f"{k}_{sfx}"] = v
            if save_params and dtype == torch.float64:
                for k, v in params.items():
                    store["param::" + k] = v.numpy()
            print(tag, sfx, "E =", res["energies"].ravel()[:4], "|grad|max =", np.abs(res["grad"]).max())
        store["in_positions"] = args[0].double().numpy()
        store["in_cells"] = args[1].double().numpy()
        store["in_centers"] = args[2].numpy().astype(np.int32)
        store["in_neighbors"] = args[3].numpy().astype(np.int32)
        store["in_cell_shifts"] = args[4].numpy().astype(np.int32)
        store["in_species"] = args[5].numpy().astype(np.int32)
        store["in_system_indices"] = args[6].numpy()
        np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **store)

    tiny = dict(hypers, d_pet=16, d_node=32, d_head=16, d_feedforward=32, num_heads=2)
    pet_case("pet_tiny", tiny, [(p64, z64, c64), (p40, z40, tri)],
             [torch.float32, torch.float64], save_params=True)
    pet_case("pet_default_box64", hypers, [(p64, z64, c64)],
             [torch.float32, torch.float64], want_features=True)
    p1k, z1k, c1k = opet.random_box(1000, seed=0)
    pet_case("pet_default_box1000", hypers, [(p1k, z1k, c1k)], [torch.float32, torch.float64])


def main_silu():
    """``activation = "SiLU"`` variant (transformer.py:32-49): E, per-atom E and dE/dR of the reference
    in fp32 / fp64 -> ``pet_silu_box64.npz``."""
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    hyp = dict(opet.DEFAULT_HYPERS, activation="SiLU")
    p64, z64, c64 = opet.random_box(64, seed=1)
    store = {}
    for dtype in (torch.float32, torch.float64):
        params = opet.synthetic_params(hyp, [1, 6, 7, 8], {"energy": 1}, 0, dtype)
        be = PETBackend(hyp, [1, 6, 7, 8])
        be.add_output("energy", {"energy": [1]})
        be = be.to(dtype).eval()
        be.load_state_dict(params, strict=True)
        assert list(be.state_dict().keys()) == list(params.keys()), "schema order"
        i, j, s, _ = onl.neighbor_list(p64.double().numpy(), c64.double().numpy(), [True] * 3, hyp["cutoff"])
        args = (p64.to(dtype), c64.to(dtype)[None], torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), z64,
                torch.zeros(64, dtype=torch.long))
        res, _ = run_reference(be, "energy", *args)
        sfx = {torch.float32: "f32", torch.float64: "f64"}[dtype]
        for k, v in res.items():
            store[f"{k}_{sfx}"] = v
        print("pet_silu_box64", sfx, "E =", res["energies"].ravel()[:4], "|grad|max =", np.abs(res["grad"]).max())
    store["in_positions"] = args[0].double().numpy()
    store["in_cells"] = args[1].double().numpy()
    store["in_centers"] = args[2].numpy().astype(np.int32)
    store["in_neighbors"] = args[3].numpy().astype(np.int32)
    store["in_cell_shifts"] = args[4].numpy().astype(np.int32)
    store["in_species"] = args[5].numpy().astype(np.int32)
    store["in_system_indices"] = args[6].numpy()
    np.savez_compressed(os.path.join(HERE, "pet_silu_box64.npz"), **store)


def _store_inputs(store, args):
    store["in_positions"] = args[0].double().numpy()
    store["in_cells"] = args[1].double().numpy()
    store["in_centers"] = args[2].numpy().astype(np.int32)
    store["in_neighbors"] = args[3].numpy().astype(np.int32)
    store["in_cell_shifts"] = args[4].numpy().astype(np.int32)
    store["in_species"] = args[5].numpy().astype(np.int32)
    store["in_system_indices"] = args[6].numpy()


def _reference_backend(PETBackend, hyp, dtype, targets=None, seed=0):
    from oracle import pet as opet

    targets = targets or {"energy": 1}
    params = opet.synthetic_params(hyp, [1, 6, 7, 8], targets, seed, dtype)
    be = PETBackend(hyp, [1, 6, 7, 8])
    for t, nprop in targets.items():
        be.add_output(t, {t: [nprop]})
    be = be.to(dtype)
    be.load_state_dict(params, strict=True)
    assert list(be.state_dict().keys()) == list(params.keys()), "schema order"
    return be, params


def main_box10000():
    """BASELINE.json's metric size: one 10 000-atom box (rho = 0.05 / A^3, seed 0), default hypers,
    synthetic weights (seed 0) -> ``pet_default_box10000.npz``:

    * ``atomic_ref_f32`` / ``grad_ref_f32``: the REFERENCE in fp32, eval mode (its own SDPA path);
    * ``atomic_ref_f64``: the REFERENCE in fp64, energies only (no autograd graph: the fp64 graph of the padded
      [N, 41, ...] layout does not fit this container's 62 GB);
    * ``atomic_f64`` / ``grad_f64``: the fp64 ORACLE (``oracle/pet.py``, unpadded CSR), asserted here to agree with
      the reference's fp64 per-atom energies to 1e-10 and with its fp32 gradient to the fp32 noise floor.

    The neighbour list is not stored (191 044 pairs): the tests rebuild it from the stored positions."""
    import time

    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    hyp = dict(opet.DEFAULT_HYPERS)
    pos, z, cell = opet.random_box(10000, seed=0)
    i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), [True] * 3, hyp["cutoff"])
    i, j, s = torch.tensor(i), torch.tensor(j), torch.tensor(s).long()
    sysidx = torch.zeros(10000, dtype=torch.long)
    store = {"in_positions": pos.numpy(), "in_species": z.numpy().astype(np.int32), "in_cell": cell.numpy(),
             "n_pairs": np.array(len(i))}
    t0 = time.time()
    be, _ = _reference_backend(PETBackend, hyp, torch.float32)
    res, _ = run_reference(be.eval(), "energy", pos, cell[None], i, j, s, z, sysidx)
    store["atomic_ref_f32"], store["grad_ref_f32"] = res["atomic"].ravel(), res["grad"]
    print("reference fp32:", time.time() - t0, "s; E =", res["energies"].ravel())
    del be, res
    t0 = time.time()
    be, p64 = _reference_backend(PETBackend, hyp, torch.float64)
    with torch.no_grad():
        b = be.eval().preprocess(pos.double(), i, j, z, cell[None].double(), s, sysidx, 1.0)
        nf, ef = be.calculate_features(b)
        pred, _, _ = be.predict(nf, ef, b, cell[None].double(), sysidx, ["energy"])
    store["atomic_ref_f64"] = pred["energy"][0].numpy().ravel()
    print("reference fp64 (energies):", time.time() - t0, "s; E =", store["atomic_ref_f64"].sum())
    del be, b, nf, ef, pred
    t0 = time.time()
    e, grad, atomic = opet.energy_and_gradient(p64, hyp, pos.double(), cell[None].double(), i, j, s, z, sysidx)
    store["atomic_f64"], store["grad_f64"] = atomic.numpy().ravel(), grad.numpy()
    print("oracle fp64:", time.time() - t0, "s; E =", float(e))
    err_e = np.abs(store["atomic_f64"] - store["atomic_ref_f64"]).max() / np.abs(store["atomic_ref_f64"]).max()
    err_g = np.abs(store["grad_f64"] - store["grad_ref_f32"]).max() / np.abs(store["grad_f64"]).max()
    print("oracle vs reference: per-atom E (fp64)", err_e, " dE/dR (fp64 oracle vs fp32 reference)", err_g)
    assert err_e < 1e-10 and err_g < 2e-5
    np.savez_compressed(os.path.join(HERE, "pet_default_box10000.npz"), **store)


def main_cosine():
    """``cutoff_function = "Cosine"`` (``pet/modules/utilities.py:25-39``): ``batch_cosine_box64.npz`` (12 ``batch_data``
    tensors) and ``pet_cosine_box64.npz`` (E, per-atom E, dE/dR in fp32 / fp64) from the reference."""
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    hyp = dict(opet.DEFAULT_HYPERS, cutoff_function="Cosine")
    p64, z64, c64 = opet.random_box(64, seed=1)
    i, j, s, _ = onl.neighbor_list(p64.double().numpy(), c64.double().numpy(), [True] * 3, hyp["cutoff"] + 1.0)
    perm = torch.randperm(len(i), generator=torch.Generator().manual_seed(3))
    i, j, s = torch.tensor(i)[perm], torch.tensor(j)[perm], torch.tensor(s).long()[perm]
    sysidx = torch.zeros(64, dtype=torch.long)
    be = PETBackend(hyp, [1, 6, 7, 8])
    batch = be.preprocess(p64, i, j, z64, c64[None], s, sysidx, 1.0)
    out = {"in_" + k: v.numpy() for k, v in dict(positions=p64, species=z64, cells=c64[None], centers=i, neighbors=j,
                                                 cell_shifts=s, system_indices=sysidx).items()}
    out.update({k: v.numpy() for k, v in batch.items()})
    np.savez(os.path.join(HERE, "batch_cosine_box64.npz"), **out)
    store = {}
    for dtype in (torch.float32, torch.float64):
        be, _ = _reference_backend(PETBackend, hyp, dtype)
        args = (p64.to(dtype), c64.to(dtype)[None], i, j, s, z64, sysidx)
        res, _ = run_reference(be.eval(), "energy", *args)
        sfx = {torch.float32: "f32", torch.float64: "f64"}[dtype]
        for k, v in res.items():
            store[f"{k}_{sfx}"] = v
        print("pet_cosine_box64", sfx, "E =", res["energies"].ravel(), "|grad|max =", np.abs(res["grad"]).max())
    _store_inputs(store, args)
    np.savez_compressed(os.path.join(HERE, "pet_cosine_box64.npz"), **store)


def main_train():
    """SURVEY §8(c) fixture (iv): ONE training step's loss and parameter gradients from the REFERENCE in ``train()``
    mode -- manual attention (``backend.py:380-384``), ``autograd.grad(E, positions, create_graph=True)``
    (``utils/output_gradient.py:34-40``), loss = MSE(E / n_atoms) + MSE(dE/dR) (``pet/trainer.py:432-452``,
    ``utils/loss.py:144-217``: mean reduction over the flattened batch, weights 1), ``loss.backward()`` -- on two
    periodic systems (64 + 40 atoms), fp64 and fp32 -> ``pet_train_two_systems.npz``: loss, energy / force loss
    terms, total gradient norm, the norm of every parameter's gradient and four full gradient tensors."""
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    hyp = dict(opet.DEFAULT_HYPERS)
    p64, z64, c64 = opet.random_box(64, seed=1)
    p40, z40, c40 = opet.random_box(40, seed=2)
    tri = c40.clone(); tri[1, 0] = 2.0; tri[2, 1] = -1.5
    systems = [(p64, z64, c64), (p40, z40, tri)]
    pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l, off = [], [], [], [], [], [], [], 0
    for k, (pos, z, cell) in enumerate(systems):
        i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), [True] * 3, hyp["cutoff"])
        pos_l.append(pos); z_l.append(z); cell_l.append(cell)
        i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off); s_l.append(torch.tensor(s))
        sys_l.append(torch.full((len(z),), k, dtype=torch.long)); off += len(z)
    pos, z, cells = torch.cat(pos_l), torch.cat(z_l), torch.stack(cell_l)
    i, j, s, sysidx = torch.cat(i_l), torch.cat(j_l), torch.cat(s_l).long(), torch.cat(sys_l)
    gen = torch.Generator().manual_seed(7)
    e_target = torch.randn(2, generator=gen, dtype=torch.float64) * 5.0
    g_target = torch.randn(len(z), 3, generator=gen, dtype=torch.float64) * 0.3
    n_atoms = torch.tensor([64.0, 40.0], dtype=torch.float64)
    store = {"e_target": e_target.numpy(), "g_target": g_target.numpy()}
    full = ["gnn_layers.1.trans.layers.0.norm_attention.weight", "gnn_layers.0.edge_embedder.weight",
            "node_last_layers.energy.0.energy.weight", "combination_norms.1.bias"]
    for dtype in (torch.float64, torch.float32):
        be, _ = _reference_backend(PETBackend, hyp, dtype)
        be.train()
        p = pos.to(dtype).clone().requires_grad_(True)
        b = be.preprocess(p, i, j, z, cells.to(dtype), s, sysidx, 1.0)
        assert b["edge_vectors"].requires_grad and be.training  # => manual attention
        nf, ef = be.calculate_features(b)
        pred, _, _ = be.predict(nf, ef, b, cells.to(dtype), sysidx, ["energy"])
        atomic = pred["energy"][0]
        energies = torch.zeros(2, 1, dtype=dtype).index_add(0, sysidx, atomic)[:, 0]
        (grad,) = torch.autograd.grad(energies.sum(), p, create_graph=True)
        loss_e = ((energies / n_atoms.to(dtype) - (e_target / n_atoms).to(dtype)) ** 2).mean()
        loss_f = ((grad - g_target.to(dtype)) ** 2).mean()
        loss = loss_e + loss_f
        loss.backward()
        named = dict(be.named_parameters())
        sfx = {torch.float32: "f32", torch.float64: "f64"}[dtype]
        store[f"loss_{sfx}"] = np.array([loss.item(), loss_e.item(), loss_f.item()])
        store[f"energies_{sfx}"] = energies.detach().numpy()
        store[f"grad_{sfx}"] = grad.detach().numpy()
        norms = np.array([named[k].grad.norm().item() for k in named])
        store[f"grad_norms_{sfx}"] = norms
        store[f"total_grad_norm_{sfx}"] = np.array(np.sqrt((norms ** 2).sum()))
        for k in full:
            store[f"dparam_{sfx}::{k}"] = named[k].grad.numpy()
        print("train", sfx, "loss", store[f"loss_{sfx}"], "total grad norm", store[f"total_grad_norm_{sfx}"])
    store["param_keys"] = np.array(list(named.keys()))
    _store_inputs(store, (pos, cells, i, j, s, z, sysidx))
    np.savez_compressed(os.path.join(HERE, "pet_train_two_systems.npz"), **store)


VARIANTS = {
    # tag: hypers that differ from the defaults (pet/documentation.py:159-259)
    "legacy": dict(normalization="LayerNorm", activation="SiLU", transformer_type="PostLN", featurizer_type="residual"),
    "layernorm": dict(normalization="LayerNorm"),
    "postln": dict(transformer_type="PostLN"),
    "residual": dict(featurizer_type="residual"),
}


def main_conditioning():
    """``system_conditioning = True`` (conditioning.py, backend.py:121-130,517-545): two systems with different total
    charge and spin multiplicity, for the feedforward and the residual featuriser -> ``pet_conditioning_<tag>.npz``
    (per-atom E, the node features of every readout layer, dE/dR, in fp32 / fp64)."""
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    p64, z64, c64 = opet.random_box(64, seed=1)
    p40, z40, c40 = opet.random_box(40, seed=2)
    pos_l, i_l, j_l, s_l, off = [], [], [], [], 0
    for pos, cell in ((p64, c64), (p40, c40)):
        i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), [True] * 3, 4.5)
        i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off); s_l.append(torch.tensor(s))
        off += len(pos)
    pos, z = torch.cat([p64, p40]), torch.cat([z64, z40])
    cells = torch.stack([c64, c40])
    i, j, s = torch.cat(i_l), torch.cat(j_l), torch.cat(s_l).long()
    sysidx = torch.cat([torch.zeros(64, dtype=torch.long), torch.ones(40, dtype=torch.long)])
    charge, spin = torch.tensor([-2, 3]), torch.tensor([1, 4])
    for tag, delta in (("feedforward", {}), ("residual", {"featurizer_type": "residual"})):
        hyp = dict(opet.DEFAULT_HYPERS, system_conditioning=True, **delta)
        store = {"in_charge": charge.numpy(), "in_spin_multiplicity": spin.numpy()}
        for dtype in (torch.float32, torch.float64):
            be, _ = _reference_backend(PETBackend, hyp, dtype)
            be = be.eval()
            p = pos.to(dtype).clone().requires_grad_(True)
            batch = be.preprocess(p, i, j, z, cells.to(dtype), s, sysidx, 1.0)
            batch["charge"], batch["spin_multiplicity"], batch["system_indices"] = charge, spin, sysidx
            nf, ef = be.calculate_features(batch)
            pred, _, _ = be.predict(nf, ef, batch, cells.to(dtype), sysidx, ["energy"])
            atomic = pred["energy"][0]
            (grad,) = torch.autograd.grad(atomic.sum(), p)
            sfx = {torch.float32: "f32", torch.float64: "f64"}[dtype]
            store[f"atomic_{sfx}"] = atomic.detach().numpy()
            store[f"grad_{sfx}"] = grad.numpy()
            if dtype == torch.float64:
                store["n_readout"] = np.array(len(nf))
                for l in range(len(nf)):
                    store[f"node_features_{l}_f64"] = nf[l].detach().numpy()
            print(tag, sfx, "E =", float(atomic.sum()), "|grad|max =", float(grad.abs().max()))
        _store_inputs(store, (pos.double(), cells.double(), i, j, s, z, sysidx))
        np.savez_compressed(os.path.join(HERE, f"pet_conditioning_{tag}.npz"), **store)


def main_multitarget():
    """Several targets, blocks and properties (backend.py:171-217, :689-777) and the non-conservative stress
    post-processing (:780-813): energy + a target with two blocks of 3 and 6 properties + a 3x3 stress target on a
    50-atom box -> ``pet_multitarget_box50.npz`` (every block's per-atom predictions, dE/dR of a weighted sum; fp64)."""
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    hyp = dict(opet.DEFAULT_HYPERS)
    targets = {"energy": 1, "multi": {"a": 3, "b": 6}, "non_conservative_stress": 9}
    params = opet.synthetic_params(hyp, [1, 6, 7, 8], targets, 0, torch.float64)
    be = PETBackend(hyp, [1, 6, 7, 8])
    be.add_output("energy", {"energy": [1]})
    be.add_output("multi", {"a": [3], "b": [3, 2]})
    be.add_output("non_conservative_stress", {"non_conservative_stress": [3, 3, 1]})
    be = be.to(torch.float64)
    be.load_state_dict(params, strict=True)
    assert list(be.state_dict().keys()) == list(params.keys()), "schema order"
    be = be.eval()
    pos, z, cell = opet.random_box(50, 9)
    i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), [True] * 3, hyp["cutoff"])
    i, j, s = torch.tensor(i), torch.tensor(j), torch.tensor(s).long()
    sysidx = torch.zeros(50, dtype=torch.long)
    cells = cell[None].double()
    p = pos.double().clone().requires_grad_(True)
    batch = be.preprocess(p, i, j, z, cells, s, sysidx, 1.0)
    nf, ef = be.calculate_features(batch)
    pred, _, _ = be.predict(nf, ef, batch, cells, sysidx, ["energy", "multi", "non_conservative_stress"])
    gen = torch.Generator().manual_seed(1)
    wa, wb = torch.randn(50, 3, generator=gen).double(), torch.randn(50, 6, generator=gen).double()
    (grad,) = torch.autograd.grad((pred["multi"][0] * wa).sum() + (pred["multi"][1] * wb).sum() + pred["energy"][0].sum(), p)
    store = {"energy": pred["energy"][0].detach().numpy(), "multi_a": pred["multi"][0].detach().numpy(),
             "multi_b": pred["multi"][1].detach().numpy(),
             "non_conservative_stress": pred["non_conservative_stress"][0].detach().numpy(),
             "wa": wa.numpy(), "wb": wb.numpy(), "grad": grad.numpy()}
    print({k: v.shape for k, v in store.items()})
    _store_inputs(store, (pos.double(), cells, i, j, s, z, sysidx))
    np.savez_compressed(os.path.join(HERE, "pet_multitarget_box50.npz"), **store)


def main_variants():
    """the variants older / production checkpoints use (pet/checkpoints.py:190-205 upgrades them to
    LayerNorm + SiLU + PostLN + residual featuriser = "legacy" here) and each switch on its own -- E, per-atom E, dE/dR
    and (legacy) the features of every readout layer, from the reference in fp32 / fp64 on the 64-atom box ->
    ``pet_variant_<tag>_box64.npz``."""
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    p64, z64, c64 = opet.random_box(64, seed=1)
    for tag, delta in VARIANTS.items():
        hyp = dict(opet.DEFAULT_HYPERS, **delta)
        i, j, s, _ = onl.neighbor_list(p64.double().numpy(), c64.double().numpy(), [True] * 3, hyp["cutoff"])
        store = {}
        for dtype in (torch.float32, torch.float64):
            be, _ = _reference_backend(PETBackend, hyp, dtype)
            args = (p64.to(dtype), c64.to(dtype)[None], torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), z64,
                    torch.zeros(64, dtype=torch.long))
            pos = args[0].clone().requires_grad_(True)
            be = be.eval()
            batch = be.preprocess(pos, args[2], args[3], args[5], args[1], args[4], args[6], 1.0)
            nf, ef = be.calculate_features(batch)
            pred, _, _ = be.predict(nf, ef, batch, args[1], args[6], ["energy"])
            atomic = pred["energy"][0]
            (grad,) = torch.autograd.grad(atomic.sum(), pos)
            sfx = {torch.float32: "f32", torch.float64: "f64"}[dtype]
            store[f"energies_{sfx}"] = atomic.sum(0, keepdim=True).detach().numpy()
            store[f"atomic_{sfx}"] = atomic.detach().numpy()
            store[f"grad_{sfx}"] = grad.numpy()
            if dtype == torch.float64:
                store["n_readout"] = np.array(len(nf))
                for l in range(len(nf)):
                    store[f"node_features_{l}_f64"] = nf[l].detach().numpy()
                    if tag == "legacy":  # [64, M, 128] per layer: kept for one variant, in fp32, to bound the fixture size
                        store[f"edge_features_{l}_f64_as_f32"] = ef[l].detach().float().numpy()
                store["padding_mask"] = batch["padding_mask"].numpy()
            print(tag, sfx, "E =", float(atomic.sum()), "|grad|max =", float(grad.abs().max()), "readout layers", len(nf))
        _store_inputs(store, args)
        np.savez_compressed(os.path.join(HERE, f"pet_variant_{tag}_box64.npz"), **store)


SIZES = {
    # tag: model sizes that differ from the defaults (pet/documentation.py:196-213); the tuned kernels of the build are ONE
    # instantiation (128 / 256 / 256 / 128 / 8), every other size runs on its size-generic path
    "s64": dict(d_pet=64, d_node=128, d_feedforward=128, d_head=64, num_heads=4),
    "flat32": dict(d_pet=32, d_node=32, d_feedforward=48, d_head=24, num_heads=2),   # d_node == d_pet: transformer.py:189-201
    "flat32_legacy": dict(d_pet=32, d_node=32, d_feedforward=48, d_head=24, num_heads=2, normalization="LayerNorm",
                          activation="SiLU", transformer_type="PostLN", featurizer_type="residual"),  # what
    # pet/checkpoints.py:190-205 upgrades pre-d_node checkpoints to
    "wide256": dict(d_pet=256, d_node=512, d_feedforward=320, d_head=96, num_heads=4),  # head dimension 64
    "minimal": dict(d_pet=1, d_node=1, d_feedforward=1, d_head=1, num_heads=1, num_attention_layers=1,
                    num_gnn_layers=1),   # pet/tests/test_basic.py:22-32 (minimal_model_hypers)
}


def main_sizes():
    """The reference at other model sizes -> ``pet_size_<tag>_box64.npz``: per-atom E, dE/dR (fp32 / fp64), the node
    features of every readout layer (fp64), on the 64-atom box with the synthetic weight generator."""
    from oracle import nl as onl
    from oracle import pet as opet

    PETBackend = import_reference_backend()
    torch.set_num_threads(8)
    p64, z64, c64 = opet.random_box(64, seed=1)
    for tag, delta in SIZES.items():
        hyp = dict(opet.DEFAULT_HYPERS, **delta)
        i, j, s, _ = onl.neighbor_list(p64.double().numpy(), c64.double().numpy(), [True] * 3, hyp["cutoff"])
        store = {}
        for dtype in (torch.float32, torch.float64):
            be, _ = _reference_backend(PETBackend, hyp, dtype)
            args = (p64.to(dtype), c64.to(dtype)[None], torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), z64,
                    torch.zeros(64, dtype=torch.long))
            pos = args[0].clone().requires_grad_(True)
            be = be.eval()
            batch = be.preprocess(pos, args[2], args[3], args[5], args[1], args[4], args[6], 1.0)
            nf, ef = be.calculate_features(batch)
            pred, _, _ = be.predict(nf, ef, batch, args[1], args[6], ["energy"])
            atomic = pred["energy"][0]
            (grad,) = torch.autograd.grad(atomic.sum(), pos)
            sfx = {torch.float32: "f32", torch.float64: "f64"}[dtype]
            store[f"energies_{sfx}"] = atomic.sum(0, keepdim=True).detach().numpy()
            store[f"atomic_{sfx}"] = atomic.detach().numpy()
            store[f"grad_{sfx}"] = grad.numpy()
            if dtype == torch.float64:
                store["n_readout"] = np.array(len(nf))
                for l in range(len(nf)):
                    store[f"node_features_{l}_f64"] = nf[l].detach().numpy()
            print(tag, sfx, "E =", float(atomic.sum()), "|grad|max =", float(grad.abs().max()), "readout layers", len(nf))
        _store_inputs(store, args)
        np.savez_compressed(os.path.join(HERE, f"pet_size_{tag}_box64.npz"), **store)


def main_soap_ps():
    """SOAP power spectrum: the REFERENCE's ``soap_bpnn/modules/power_spectrum.py`` imported unchanged and run on the
    oracle's spherical expansion. torch-spex (the expansion itself) is not installable here, so ``spex.spherical_expansion.
    SphericalExpansion`` is a stand-in that returns ``oracle.soap.spherical_expansion`` in spex's output format (a list over
    l of ``[centre, 2l+1, n_l, channel]``); everything downstream of it -- the per-l ``einsum("smn,smN->snN")`` contraction,
    the flattening and concatenation order over (l, n, channel, n', channel'), the split into ``center_type`` blocks of the
    legacy model -- is reference code. Writes soap_ps_box24.npz (legacy / Orthogonal and Alchemical)."""
    from oracle import nl as onl
    from oracle import pet as opet
    from oracle import soap as osoap

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Labels:
        def __init__(self, names, values):
            self.names, self.values = names, values

        def to(self, device):
            return self

    class TensorBlock:
        def __init__(self, values, samples, components, properties):
            self.values, self.samples, self.components, self.properties = values, samples, components, properties

    class TensorMap:
        def __init__(self, keys, blocks):
            self.keys, self.blocks = keys, blocks

    state = {}

    class _Radial:
        pass

    class SphericalExpansion:  # spex's constructor signature as the reference calls it (power_spectrum.py:42)
        def __init__(self, cutoff, max_angular, radial, angular, species, cutoff_function):
            self.max_angular = max_angular
            self.radial = _Radial()
            self.radial.n_per_l = state["n_per_l"]

        def forward(self, R_ij, i, j, species):
            return osoap.spherical_expansion(R_ij, i, state["sp_index"][j], len(species), state["hypers"], state["weights"])

    stub("metatensor")
    stub("metatensor.torch", Labels=Labels, TensorBlock=TensorBlock, TensorMap=TensorMap)
    stub("spex")
    stub("spex.spherical_expansion", SphericalExpansion=SphericalExpansion)
    spec = importlib.util.spec_from_file_location("ref_power_spectrum", f"{REF}/soap_bpnn/modules/power_spectrum.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    atomic_types = [1, 6, 7, 8]
    n_at = 24  # 4 544 features per atom: kept small (the file holds two [n_at, 4544] fp32 arrays)
    pos, z, cell = opet.random_box(n_at, seed=21)
    out = {"in_positions": pos.numpy(), "in_species": z.numpy(), "in_cell": cell.numpy()}
    for tag, legacy in (("legacy", True), ("alchemical", False)):
        hypers = dict(osoap.DEFAULT_HYPERS, legacy=legacy)
        so = hypers["soap"]
        n_per_l = osoap.basis(hypers)[0]
        i, j, sh, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, float(so["cutoff"]["radius"]))
        i, j, sh = torch.tensor(i), torch.tensor(j), torch.tensor(sh)
        p64 = pos.double()
        v = p64[j] - p64[i] + sh.double() @ cell.double()
        table = torch.full((max(atomic_types) + 1,), -1, dtype=torch.long)
        table[torch.tensor(atomic_types)] = torch.arange(len(atomic_types))
        params = osoap.synthetic_params(hypers, len(atomic_types), n_per_l, 3, torch.float64)
        weights = torch.eye(4, dtype=torch.float64) if legacy else params["species_embedding.weight"].double()
        state.update(n_per_l=n_per_l, hypers=hypers, weights=weights, sp_index=table[z.long()])
        species_spec = ({"Orthogonal": {"species": atomic_types}} if legacy
                        else {"Alchemical": {"pseudo_species": 4, "total_species": len(atomic_types)}})
        ps = mod.SoapPowerSpectrum(float(so["cutoff"]["radius"]), so["max_angular"],
                                   {"LaplacianEigenstates": {"max_radial": so["max_radial"]}}, "SphericalHarmonics",
                                   species_spec, {"ShiftedCosine": {"width": so["cutoff"]["width"]}})
        tmap = ps.forward(v, i, j, z.long(), torch.zeros(n_at, dtype=torch.long), torch.arange(n_at))
        full = torch.zeros((n_at, ps.shape), dtype=torch.float64)
        if legacy:
            keys = tmap.keys.values.reshape(-1).tolist()
            for key, block in zip(keys, tmap.blocks):
                atoms = block.samples.values[:, 1]
                assert bool((z[atoms].long() == key).all())
                full[atoms] = block.values
            out["legacy_center_types"] = np.array(keys)
        else:
            (block,) = tmap.blocks
            full[block.samples.values[:, 1]] = block.values
            out["alchemical_species_embedding"] = weights.numpy()
        assert full.shape[1] == osoap.soap_size(n_per_l, 4)
        out[f"{tag}_power_spectrum"] = full.numpy().astype(np.float32)
        out[f"{tag}_pairs"] = torch.cat([i[:, None], j[:, None], sh], 1).numpy()
        print(tag, "power spectrum", tuple(full.shape), "|max|", float(full.abs().max()))
    np.savez_compressed(os.path.join(HERE, "soap_ps_box24.npz"), **out)


if __name__ == "__main__":
    if "--soap-ps" in sys.argv:
        main_soap_ps()
    elif "--sizes" in sys.argv:
        main_sizes()
    elif "--multitarget" in sys.argv:
        main_multitarget()
    elif "--conditioning" in sys.argv:
        main_conditioning()
    elif "--variants" in sys.argv:
        main_variants()
    elif "--box10000" in sys.argv:
        main_box10000()
    elif "--cosine" in sys.argv:
        main_cosine()
    elif "--train" in sys.argv:
        main_train()
    elif "--silu" in sys.argv:
        main_silu()
    elif "--adaptive-grid" in sys.argv:
        main_adaptive("grid")
    elif "--adaptive" in sys.argv:
        main_adaptive()
    else:
        main()

"""Cases that sit ON the fp32 floor (round-2 fuzz sweeps, ``tests/debug/fuzz_parity.py``; VERDICT r2 hygiene: bring them into
the suite): a steep cutoff flank (``cutoff_function = "Cosine"``, ``cutoff_width = 0.2``, 1 GNN x 3 attention layers, SiLU) and
very dilute batches, where torch's own fp32 evaluation of the SAME model and inputs misses its fp64 values by about 1e-5.
Bound for the HIP path: the 1e-5 bar, or 1.5 x that fp32 yardstick where the yardstick itself is above the bar."""
import numpy as np
import pytest
import torch

from oracle import nl as onl
from oracle import pet as opet

from _memo import memo_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5
TYPES = [1, 6, 7, 8]


def _batch(rng, dilute):
    n_sys = int(rng.integers(1, 4))
    pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l, off = [], [], [], [], [], [], [], 0
    for k in range(n_sys):
        n = int(rng.integers(20, 160))
        rho = float(10 ** rng.uniform(-2.6, -2.2)) if dilute else float(10 ** rng.uniform(-1.6, -1.0))
        box = max((n / rho) ** (1 / 3), 3.0)
        cell = np.eye(3) * box + (rng.uniform(-0.25, 0.25, (3, 3)) * box if rng.random() < 0.5 else 0.0)
        pos = rng.random((n, 3)) @ cell
        return_pbc = [True, True, True]
        i, j, s, _ = onl.neighbor_list(pos, cell, return_pbc, 5.5)
        pos_l.append(torch.tensor(pos)); z_l.append(torch.tensor(rng.choice(TYPES, n)).int()); cell_l.append(torch.tensor(cell))
        i_l.append(torch.tensor(i) + off); j_l.append(torch.tensor(j) + off); s_l.append(torch.tensor(s).long())
        sys_l.append(torch.full((n,), k, dtype=torch.long))
        off += n
    return (torch.cat(pos_l), torch.stack(cell_l), torch.cat(i_l), torch.cat(j_l), torch.cat(s_l), torch.cat(z_l), torch.cat(sys_l))


@memo_oracle
def _oracle(params, hypers, b, dtype):
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in params.items()}
    e, g, a = opet.energy_and_gradient(p, hypers, b[0].to(dtype), b[1].to(dtype), b[2], b[3], b[4], b[5], b[6])
    return a.double().numpy().ravel(), g.double().numpy()


@pytest.mark.parametrize("case", ["steep_cosine_flank", "dilute"])
def test_hip_error_against_the_fp32_yardstick(case):
    from metatrain_amd import runtime as rt

    dev = torch.device("cuda:0")
    hypers = dict(opet.DEFAULT_HYPERS)
    if case == "steep_cosine_flank":
        hypers.update(cutoff_function="Cosine", cutoff_width=0.2, num_gnn_layers=1, num_attention_layers=3, activation="SiLU",
                      cutoff=5.5)
    else:
        hypers.update(cutoff=5.5)
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    model = rt.HipModel(hypers, TYPES)
    model.load({k: v.to(dev) for k, v in params.items()}, "energy")
    rng = np.random.default_rng(101 if case == "steep_cosine_flank" else 7)
    worst = []
    for _ in range(4):
        b = _batch(rng, dilute=(case == "dilute"))
        if len(b[2]) == 0:
            continue
        a64, g64 = _oracle(params, hypers, b, torch.float64)
        a32, g32 = _oracle(params, hypers, b, torch.float32)
        graph = rt.HipGraph(model, b[0].float().to(dev), b[1].float().to(dev), b[2].int().to(dev), b[3].int().to(dev),
                            b[4].int().to(dev), b[5].to(dev), b[6].int().to(dev))
        fw = rt.HipForward(model, graph)
        a = fw.forward().cpu().numpy().astype(np.float64)
        g = fw.backward(torch.ones(len(a), device=dev)).cpu().numpy().astype(np.float64)
        rel = lambda x, r: np.abs(x - r).max() / np.abs(r).max()  # noqa: E731
        yard_e, yard_g = rel(a32, a64), rel(g32, g64)
        err_e, err_g = rel(a, a64), rel(g, g64)
        worst.append((err_e, yard_e, err_g, yard_g))
        assert err_e < max(TOL, 1.5 * yard_e), (err_e, yard_e)
        assert err_g < max(TOL, 1.5 * yard_g), (err_g, yard_g)
    print(case, "HIP / torch-fp32 errors (E, dE/dR):", [(f"{a:.1e}", f"{b:.1e}", f"{c:.1e}", f"{d:.1e}") for a, b, c, d in worst])


def _dilute_partly_periodic_box(seed, cutoff):
    """The regime of the round-5 sweep outliers (profiles/r05_fuzz_summary.txt): ONE very dilute system (0.0025 - 0.005 atoms
    per cubic Angstrom: a 30 - 40 A box), possibly triclinic, with at least one non-periodic direction."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(90, 200))
    rho = float(10 ** rng.uniform(-2.6, -2.3))
    box = (n / rho) ** (1 / 3)
    cell = np.eye(3) * box + (rng.uniform(-0.25, 0.25, (3, 3)) * box if rng.random() < 0.5 else 0.0)
    pbc = [bool(b) for b in rng.random(3) < 0.7]
    pbc[int(rng.integers(0, 3))] = False
    if not any(pbc):
        pbc[int(rng.integers(0, 3))] = True
    pos = rng.random((n, 3)) @ cell
    i, j, s, _ = onl.neighbor_list(pos, cell, pbc, cutoff)
    b = (torch.tensor(pos), torch.tensor(cell)[None], torch.tensor(i, dtype=torch.int64), torch.tensor(j, dtype=torch.int64),
         torch.tensor(s, dtype=torch.int64).reshape(-1, 3), torch.tensor(rng.choice(TYPES, n)).int(), torch.zeros(n, dtype=torch.long))
    return b, torch.tensor(rng.uniform(0.2, 2.0, n))


@memo_oracle
def _oracle_with_cell(params, hypers, b, w, dtype):
    p = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in params.items()}
    q = b[0].float().to(dtype).requires_grad_(True)     # (both precisions start from the fp32-rounded inputs the GPU receives)
    c = b[1].float().to(dtype).requires_grad_(True)
    a = opet.pet_atomic_energies(p, hypers, q, c, b[2], b[3], b[4], b[5], b[6]).ravel()
    gp, gc = torch.autograd.grad((a * w.to(dtype)).sum(), [q, c], allow_unused=True)
    gc = torch.zeros_like(c) if gc is None else gc
    return a.detach().double().numpy(), gp.double().numpy(), gc.double().numpy()


def test_dilute_partly_periodic_boxes_against_the_fp32_yardstick():
    """VERDICT r5 item 7a. The round-5 sweeps flagged dilute, partly periodic batches with the adaptive cutoff at up to 1.15e-5
    (dE/dR) and 5.4e-5 (dE/dcell, relative to its largest component) of the fp64 oracle. torch's own fp32 evaluation of the
    SAME model on the SAME inputs loses the same amounts (round 6, tools/debug/cell_outliers.sh, HIP / torch fp32: 1.15e-5 /
    1.15e-5, 5.40e-5 / 5.28e-5, 3.41e-5 / 3.44e-5, 1.96e-5 / 1.96e-5): edge vectors of a 30 - 40 A box are differences of
    numbers ten times their size, and dE/dcell of a box with a handful of boundary-crossing edges is a small sum (1e-2 .. 1e-1 of
    the largest force) of such terms. Five seeded boxes of that regime -- the yardstick itself is above 1e-5 in dE/dR on two of
    them and in dE/dcell on three -- with the round-5 kernels and the fused attention block forced on these small graphs: the
    HIP path stays within 1e-5, or within 2 x the yardstick where that is larger, in E, dE/dR and dE/dcell."""
    from metatrain_amd import runtime as rt

    dev = torch.device("cuda:0")
    hypers = dict(opet.DEFAULT_HYPERS)
    hypers.update(num_neighbors_adaptive=10, adaptive_cutoff_method="solver", cutoff_width_adaptive=1.0)
    params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
    rt.config_set("emlp_s", 2)
    rt.config_set("attn_fused", 7)
    rows = []
    try:
        model = rt.HipModel(hypers, TYPES)
        model.load({k: v.to(dev) for k, v in params.items()}, "energy")
        for seed in (101, 114, 123, 125, 127):
            b, w = _dilute_partly_periodic_box(seed, hypers["cutoff"])
            a64, g64, c64 = _oracle_with_cell(params, hypers, b, w, torch.float64)
            a32, g32, c32 = _oracle_with_cell(params, hypers, b, w, torch.float32)
            graph = rt.HipGraph(model, b[0].float().to(dev), b[1].float().to(dev), b[2].to(dev), b[3].to(dev), b[4].to(dev),
                                b[5].to(dev), b[6].int().to(dev))
            fw = rt.HipForward(model, graph)
            a = fw.forward().cpu().numpy().astype(np.float64).ravel()
            g, gc = fw.backward(w.float().to(dev), want_cell_grad=True)
            g, gc = g.cpu().numpy().astype(np.float64), gc.cpu().numpy().astype(np.float64)
            rel = lambda x, r: float(np.abs(x - r).max() / max(np.abs(r).max(), 1e-30))  # noqa: E731
            errs = (rel(a, a64), rel(g, g64), rel(gc, c64))
            yard = (rel(a32, a64), rel(g32, g64), rel(c32, c64))
            rows.append((errs, yard))
            for name, e, y in zip(("E", "dE/dR", "dE/dcell"), errs, yard):
                assert e < max(TOL, 2.0 * y), (seed, name, e, y)
    finally:
        rt.config_set("emlp_s", 1)
        rt.config_set("attn_fused", 3)
    print("HIP / torch-fp32 errors (E, dE/dR, dE/dcell):", [tuple(f"{e:.1e}/{y:.1e}" for e, y in zip(er, ya)) for er, ya in rows])
    # the regime the test is about: the yardstick itself is above the bar on several of these boxes
    assert sum(ya[1] > TOL for _, ya in rows) >= 2 and sum(ya[2] > TOL for _, ya in rows) >= 3

"""Fused attention block (csrc/pet_ablk.hip) against the three-kernel form on one GPU box: per-atom energies (forward) and
dE/dR (adjoint) of both against each other and against the reference's fp64 goldens, then per-stage times at
BOXES x 10k atoms. MODE = 1 fused forward only (energies), 3 = forward + adjoint."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params

MODE = int(os.environ.get("MODE", "1"))
dev = torch.device("cuda:0")
hypers = default_hypers()
if os.environ.get("NORM"):
    hypers["normalization"] = os.environ["NORM"]
params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")
G = os.path.join(ROOT, "tests", "golden")
rel = lambda x, y: float(np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)).max() / np.abs(np.asarray(y, np.float64)).max())


def golden(name):
    g = dict(np.load(os.path.join(G, name)))
    t = lambda k: torch.tensor(g[k]).to(dev)
    if "in_cells" in g:
        graph = rt.HipGraph(model, t("in_positions").float(), t("in_cells").float(), t("in_centers"), t("in_neighbors"),
                            t("in_cell_shifts"), t("in_species"), t("in_system_indices").int())
    else:  # the 10 000-atom golden stores positions / species / cell only; the list is built on the device
        pos, cell = t("in_positions").float(), torch.tensor(g["in_cell"]).float()
        pr, _ = rt.neighbor_list(pos, cell, [True] * 3, 4.5)
        graph = rt.HipGraph(model, pos, cell[None].to(dev), pr[:, 0].contiguous(), pr[:, 1].contiguous(),
                            pr[:, 2:5].contiguous(), t("in_species"), torch.zeros(len(pos), dtype=torch.int32, device=dev))
    out = {}
    for mode in (0, MODE):
        rt.config_set("attn_fused", mode)
        fw = rt.HipForward(model, graph)
        a = fw.forward()
        gp = fw.backward(torch.ones_like(a)) if (mode == 0 or mode & 2) else None
        torch.cuda.synchronize()
        out[mode] = (a.cpu().numpy(), None if gp is None else gp.cpu().numpy())
    a0, g0 = out[0]
    a1, g1 = out[MODE]
    msg = f"{name}: atomic fused-vs-old {rel(a1, a0):.2e}"
    if hypers["normalization"] == "RMSNorm":
        msg += f" | vs fp64: old {rel(a0, g['atomic_f64'].ravel()):.2e} fused {rel(a1, g['atomic_f64'].ravel()):.2e}"
    if g1 is not None:
        msg += f" || grad fused-vs-old {rel(g1, g0):.2e}"
        if hypers["normalization"] == "RMSNorm":
            msg += f" | vs fp64: old {rel(g0, g['grad_f64']):.2e} fused {rel(g1, g['grad_f64']):.2e}"
    print(msg, flush=True)


for n in ("pet_default_box64.npz", "pet_default_box1000.npz", "pet_default_box10000.npz"):
    golden(n)

# a dense box: atoms of 33 .. 64 tokens (the NQ = 2 instantiation)
pos, z, cell = random_box(1500, 5, density=0.095)
pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5)
graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                    pairs[:, 2:5].contiguous(), z.to(dev), torch.zeros(len(z), dtype=torch.int32, device=dev))
res = {}
for mode in (0, MODE):
    rt.config_set("attn_fused", mode | (4 if mode else 0))  # 4: fused although most atoms take the 64-slot tiles
    fw = rt.HipForward(model, graph)
    a = fw.forward()
    gp = fw.backward(torch.ones_like(a)) if (mode == 0 or mode & 2) else None
    torch.cuda.synchronize()
    res[mode] = (a.cpu().numpy(), None if gp is None else gp.cpu().numpy())
print(f"dense box: {pairs.shape[0] / len(z):.1f} neighbours/atom, atomic fused-vs-old {rel(res[MODE][0], res[0][0]):.2e}"
      + ("" if res[MODE][1] is None else f" grad {rel(res[MODE][1], res[0][1]):.2e}"), flush=True)

nb = int(os.environ.get("BOXES", "8"))
P, Z, C, PR, S = [], [], [], [], []
for b in range(nb):
    pos, z, cell = random_box(10000, b)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5); pairs = pairs.clone(); pairs[:, :2] += b * 10000
    P.append(pos.to(dev)); Z.append(z.to(dev)); C.append(cell.to(dev)); PR.append(pairs)
    S.append(torch.full((10000,), b, dtype=torch.int32, device=dev))
P, Z, C, PR, S = torch.cat(P), torch.cat(Z), torch.stack(C), torch.cat(PR), torch.cat(S)
ones = torch.ones(nb * 10000, device=dev)
for mode in (0, MODE):
    rt.config_set("attn_fused", mode)
    st = {}

    def step():
        graph = rt.HipGraph(model, P, C, PR[:, 0].contiguous(), PR[:, 1].contiguous(), PR[:, 2:5].contiguous(), Z, S)
        if "fw" not in st:
            st["fw"] = rt.HipForward(model, graph)
        st["fw"].graph = graph
        a = st["fw"].forward()
        return a, st["fw"].backward(ones)

    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        a, gp = step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(f"attn_fused={mode}: {nb}x10k atoms: {dt*1e3:.2f} ms/step -> {nb*10000/dt:.0f} atom-steps/s", flush=True)
    rt.config_set("side_stream", 0)
    rt.profile(True); step(); torch.cuda.synchronize(); rep = rt.profile_report(); rt.profile(False)
    rt.config_set("side_stream", 1)
    for r in sorted(rep, key=lambda r: -r["total_ms"])[:14]:
        print(f"  {r['name']:16s} {r['total_ms']:8.3f} ms  x{r['calls']}", flush=True)

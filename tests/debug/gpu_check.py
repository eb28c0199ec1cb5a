"""Diagnostic run on a GPU box: HIP path vs golden fixtures / oracle, stage by stage."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metatrain_amd import runtime as rt
from oracle import pet as opet, nl as onl

dev = torch.device("cuda:0")
G = os.path.join(ROOT, "tests", "golden")

def relmax(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)

hypers = dict(opet.DEFAULT_HYPERS)
params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in params.items()}, "energy")
print("model params", model.num_params)

# ---- batch goldens
for case in ["co2cell", "box64", "two_systems"]:
    g = dict(np.load(os.path.join(G, f"batch_{case}.npz")))
    t = lambda k, dt=None: torch.tensor(g[k]).to(dev) if dt is None else torch.tensor(g[k]).to(dev).to(dt)
    graph = rt.HipGraph(model, t("in_positions"), t("in_cells"), t("in_centers"), t("in_neighbors"),
                        t("in_cell_shifts"), t("in_species"), t("in_system_indices"))
    out = graph.export_batch()
    for k, v in out.items():
        ref = g[k]
        got = v.cpu().numpy()
        if got.shape != ref.shape:
            print(case, k, "SHAPE", got.shape, ref.shape); continue
        if ref.dtype.kind in "iub":
            print(case, k, "exact" if np.array_equal(got, ref) else f"MISMATCH {np.sum(got != ref)}")
        else:
            print(case, k, "maxabs", np.abs(got - ref).max())

# ---- energies / features on box64
g = dict(np.load(os.path.join(G, "pet_default_box64.npz")))
t = lambda k: torch.tensor(g[k]).to(dev)
graph = rt.HipGraph(model, t("in_positions").float(), t("in_cells").float(), t("in_centers"), t("in_neighbors"),
                    t("in_cell_shifts"), t("in_species"), t("in_system_indices").int())
fw = rt.HipForward(model, graph)
atomic, nf, ef = fw.forward(want_features=True)
torch.cuda.synchronize()
print("box64 E hip", float(atomic.sum()), "ref32", g["energies_f32"].ravel(), "ref64", g["energies_f64"].ravel())
print("atomic rel vs f64", relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()), "ref f32 vs f64", relmax(g["atomic_f32"], g["atomic_f64"]))
print("node feat rel vs f64", relmax(nf.cpu().numpy(), g["node_features_f64"]), " ref32:", relmax(g["node_features_f32"], g["node_features_f64"]))
# edge features: golden is NEF [N,M,128]; compare real slots
csr = graph.csr(); rowptr = csr["rowptr"].cpu().numpy()
efn = ef.cpu().numpy(); ref = g["edge_features_f64"]
errs = []
for i in range(graph.n_nodes):
    n = rowptr[i+1] - rowptr[i]
    errs.append(np.abs(efn[rowptr[i]:rowptr[i+1]] - ref[i, :n]).max() if n else 0.0)
print("edge feat maxabs vs f64", max(errs), "scale", np.abs(ref).max())
gpos = fw.backward(torch.ones_like(atomic))
torch.cuda.synchronize()
print("grad rel(max) vs f64", relmax(gpos.cpu().numpy(), g["grad_f64"]), " ref32 vs f64:", relmax(g["grad_f32"], g["grad_f64"]))

# ---- box1000
g = dict(np.load(os.path.join(G, "pet_default_box1000.npz")))
graph = rt.HipGraph(model, t("in_positions").float(), t("in_cells").float(), t("in_centers"), t("in_neighbors"),
                    t("in_cell_shifts"), t("in_species"), t("in_system_indices").int())
fw = rt.HipForward(model, graph)
atomic = fw.forward(); gpos = fw.backward(torch.ones_like(atomic)); torch.cuda.synchronize()
print("box1000 E hip", float(atomic.double().sum()), "ref32", g["energies_f32"].ravel(), "ref64", g["energies_f64"].ravel())
print("box1000 atomic rel", relmax(atomic.cpu().numpy(), g["atomic_f64"].ravel()), "grad rel", relmax(gpos.cpu().numpy(), g["grad_f64"]), "ref32:", relmax(g["grad_f32"], g["grad_f64"]))

# ---- NL
pos, z, cell = opet.random_box(1000, 5)
pairs, vec = rt.neighbor_list(pos.to(dev), cell, [True]*3, 4.5)
i, j, s, d = onl.neighbor_list(pos.numpy(), cell.numpy(), [True]*3, 4.5)
got = pairs.cpu().numpy(); got = got[np.lexsort((got[:,4], got[:,3], got[:,2], got[:,1], got[:,0]))]
ref = np.column_stack([i, j, s])
print("NL pairs hip", len(got), "oracle", len(ref), "equal sets:", got.shape == ref.shape and np.array_equal(got, ref))

# ---- timing at 10k atoms
pos, z, cell = opet.random_box(10000, 0)
posd = pos.to(dev)
t0 = time.time(); pairs, vec = rt.neighbor_list(posd, cell, [True]*3, 4.5); torch.cuda.synchronize(); print("NL 10k", time.time()-t0, "s", len(pairs))
sysidx = torch.zeros(10000, dtype=torch.int32, device=dev)
cells = cell[None].to(dev)
def step():
    graph = rt.HipGraph(model, posd, cells, pairs[:,0], pairs[:,1], pairs[:,2:5], z.to(dev), sysidx)
    fw = rt.HipForward(model, graph)
    a = fw.forward(); gp = fw.backward(torch.ones_like(a)); return a, gp
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): a, gp = step()
torch.cuda.synchronize(); dt = (time.time()-t0)/5
print(f"10k atoms fwd+bwd {dt*1e3:.2f} ms/step -> {10000/dt:.0f} atom-steps/s; E={float(a.sum()):.4f}")
rt.profile(True); step(); torch.cuda.synchronize()
rep = rt.profile_report(); rt.profile(False)
for r in sorted(rep, key=lambda r: -r["total_ms"]):
    print(f"  {r['name']:16s} {r['total_ms']:8.3f} ms  x{r['calls']}  {r['flops']/max(r['total_ms'],1e-9)/1e9:8.1f} TF/s")

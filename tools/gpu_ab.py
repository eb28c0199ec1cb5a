"""A/B on one GPU box: per-stage times at 2 x 10k atoms (run with PET_HIP_TRR=0/1) + parity."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params
dev = torch.device("cuda:0")
hypers = default_hypers()
params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")
G = os.path.join(ROOT, "tests", "golden")
g = dict(np.load(os.path.join(G, "pet_default_box1000.npz")))
t = lambda k: torch.tensor(g[k]).to(dev)
graph = rt.HipGraph(model, t("in_positions").float(), t("in_cells").float(), t("in_centers"), t("in_neighbors"), t("in_cell_shifts"), t("in_species"), t("in_system_indices").int())
fw = rt.HipForward(model, graph); a = fw.forward(); gp = fw.backward(torch.ones_like(a)); torch.cuda.synchronize()
rel = lambda x, y: np.abs(np.asarray(x, np.float64) - y).max() / np.abs(y).max()
print("TRR =", os.environ.get("PET_HIP_TRR", "1"), "box1000 atomic rel", rel(a.cpu().numpy(), g["atomic_f64"].ravel()), "grad rel", rel(gp.cpu().numpy(), g["grad_f64"]))
nb = int(os.environ.get("BOXES", "2"))
P, Z, C, PR, S = [], [], [], [], []
for b in range(nb):
    pos, z, cell = random_box(10000, b)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True]*3, 4.5); pairs = pairs.clone(); pairs[:, :2] += b * 10000
    P.append(pos.to(dev)); Z.append(z.to(dev)); C.append(cell.to(dev)); PR.append(pairs); S.append(torch.full((10000,), b, dtype=torch.int32, device=dev))
P, Z, C, PR, S = torch.cat(P), torch.cat(Z), torch.stack(C), torch.cat(PR), torch.cat(S)
ones = torch.ones(nb * 10000, device=dev)
st = {}
def step():
    graph = rt.HipGraph(model, P, C, PR[:, 0].contiguous(), PR[:, 1].contiguous(), PR[:, 2:5].contiguous(), Z, S)
    if "fw" not in st: st["fw"] = rt.HipForward(model, graph)
    st["fw"].graph = graph
    a = st["fw"].forward(); return a, st["fw"].backward(ones)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): a, gp = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f"{nb}x10k atoms: {dt*1e3:.2f} ms/step -> {nb*10000/dt:.0f} atom-steps/s")
rt.profile(True); step(); torch.cuda.synchronize(); rep = rt.profile_report(); rt.profile(False)
for r in sorted(rep, key=lambda r: -r["total_ms"]):
    print(f"  {r['name']:16s} {r['total_ms']:8.3f} ms  x{r['calls']}  {r['flops']/max(r['total_ms'],1e-9)/1e9:8.1f} TF/s")

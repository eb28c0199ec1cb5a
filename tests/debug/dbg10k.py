"""Localise a parity failure at 10k atoms: per-switch errors against the golden."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metatrain_amd import runtime as rt
from oracle import nl as onl, pet as opet

dev = torch.device("cuda:0")
g = dict(np.load("tests/golden/pet_default_box10000.npz"))
hypers = dict(opet.DEFAULT_HYPERS)
params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")
pos, z, cell = torch.tensor(g["in_positions"]), torch.tensor(g["in_species"]), torch.tensor(g["in_cell"])
i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell.double().numpy(), [True] * 3, 4.5)
sysidx = torch.zeros(10000, dtype=torch.int32)
graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), torch.tensor(i).to(dev), torch.tensor(j).to(dev), torch.tensor(s).to(dev), z.to(dev), sysidx.to(dev))
ref = g["grad_f64"]; scale = np.abs(ref).max()
def run(tag):
    fw = rt.HipForward(model, graph)
    a = fw.forward(); gr = fw.backward(torch.ones_like(a)).cpu().numpy()
    err = np.abs(gr - ref).max(1) / scale
    bad = np.nonzero(err > 1e-5)[0]
    print(f"{tag:24s} E err {np.abs(a.cpu().numpy()-g['atomic_f64']).max()/np.abs(g['atomic_f64']).max():.2e}  grad max err {err.max():.2e}  atoms>1e-5: {len(bad)}  first: {bad[:8]}", flush=True)
    return gr
g0 = run("default")
g1 = run("default again")
print("run-to-run identical:", np.array_equal(g0, g1))
for sw, val, dflt in [("attn_lds", 1, 3), ("trr", 0, 1), ("side_stream", 0, 1), ("trr_compress", 0, 3), ("tile_f16x3", 0, 1)]:
    rt.config_set(sw, val)
    try:
        run(f"{sw}={val}")
    finally:
        rt.config_set(sw, dflt)
counts = np.bincount(i, minlength=10000)
print("max neighbours", counts.max(), "hist>32:", (counts > 32).sum(), ">47", (counts>47).sum())

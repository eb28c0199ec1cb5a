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
lib = rt._lib.load()
N, E = graph.n_nodes, graph.n_edges
def staged(tag):
    fw = rt.HipForward(model, graph)
    a = fw.forward()
    ga = torch.ones_like(a)
    g_nf = torch.empty((N, 256), device=dev); g_ef = torch.zeros((E, 128), device=dev); g_fc = torch.zeros(E, device=dev)
    rt.check(lib.pet_backward_predict(model.handle, graph.handle, rt._ptr(fw.workspace), fw.nbytes, rt._ptr(ga), rt._ptr(g_nf), rt._ptr(g_ef), rt._ptr(g_fc), rt._stream()))
    geo = torch.empty((E, 4), device=dev); gfc2 = torch.empty(E, device=dev)
    rt.check(lib.pet_backward_features(model.handle, graph.handle, rt._ptr(fw.workspace), fw.nbytes, rt._ptr(g_nf), rt._ptr(g_ef), rt._ptr(geo), rt._ptr(gfc2), rt._stream()))
    torch.cuda.synchronize()
    return dict(g_nf=g_nf.clone(), g_ef=g_ef.clone(), g_fc=g_fc.clone(), geo=geo.clone(), gfc2=gfc2.clone())
def full(tag):
    fw = rt.HipForward(model, graph)
    a = fw.forward(); gr = fw.backward(torch.ones_like(a)).cpu().numpy()
    err = np.abs(gr - ref).max(1) / scale
    print(f"{tag:24s} grad max err {err.max():.2e}  atoms>1e-5: {(err > 1e-5).sum()}", flush=True)
for v in (3, 1, 2, 0):
    rt.config_set("trr_compress", v)
    for rep in range(2):
        full(f"trr_compress={v} rep{rep}")
rt.config_set("trr_compress", 0)
base = staged("0")
for v in (1, 2):
    rt.config_set("trr_compress", v)
    for rep in range(3):
        cur = staged(str(v))
        msg = []
        for k in base:
            d = (cur[k] - base[k]).abs()
            sc = base[k].abs().max()
            rows = (d.reshape(d.shape[0], -1).max(1).values > 1e-4 * sc).nonzero().flatten()
            msg.append(f"{k}: {float(d.max()/sc):.1e} rows {len(rows)} {rows[:6].tolist()}")
        print(f"trr_compress={v} rep{rep} vs 0 | " + " | ".join(msg), flush=True)
rt.config_set("trr_compress", 3)

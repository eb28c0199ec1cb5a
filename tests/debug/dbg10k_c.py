import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metatrain_amd import runtime as rt
from oracle import nl as onl, pet as opet

dev = torch.device("cuda:0")
hypers = dict(opet.DEFAULT_HYPERS)
params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")
lib = rt._lib.load()
def mk(n):
    pos, z, cell = opet.random_box(n, 0)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True]*3, 4.5)
    return rt.HipGraph(model, pos.to(dev), cell[None].to(dev), pairs[:,0].contiguous(), pairs[:,1].contiguous(), pairs[:,2:5].contiguous(), z.to(dev), torch.zeros(n, dtype=torch.int32, device=dev))
def staged(graph):
    N, E = graph.n_nodes, graph.n_edges
    fw = rt.HipForward(model, graph)
    a = fw.forward()
    ga = torch.ones_like(a)
    g_nf = torch.empty((N, 256), device=dev); g_ef = torch.zeros((E, 128), device=dev); g_fc = torch.zeros(E, device=dev)
    rt.check(lib.pet_backward_predict(model.handle, graph.handle, rt._ptr(fw.workspace), fw.nbytes, rt._ptr(ga), rt._ptr(g_nf), rt._ptr(g_ef), rt._ptr(g_fc), rt._stream()))
    geo = torch.empty((E, 4), device=dev); gfc2 = torch.empty(E, device=dev)
    rt.check(lib.pet_backward_features(model.handle, graph.handle, rt._ptr(fw.workspace), fw.nbytes, rt._ptr(g_nf), rt._ptr(g_ef), rt._ptr(geo), rt._ptr(gfc2), rt._stream()))
    torch.cuda.synchronize()
    return geo.clone()
for n in (3000, 3500, 4000, 6000, 10000, 20000):
    graph = mk(n)
    rt.config_set("trr_compress", 0); base = staged(graph)
    rt.config_set("trr_compress", 3)
    for rep in range(2):
        cur = staged(graph)
        d = (cur - base).abs(); sc = float(base.abs().max())
        bad = (d.max(1).values > 1e-4 * sc).nonzero().flatten().cpu().numpy()
        if len(bad) == 0:
            print(f"n={n} E={graph.n_edges} rep{rep}: no bad rows", flush=True); continue
        cols = (d[bad] > 1e-4 * sc).sum(0).tolist()
        wg = bad // 128
        print(f"n={n} E={graph.n_edges} rep{rep}: bad rows {len(bad)} min {bad.min()} max {bad.max()} distinct WGs {len(np.unique(wg))} "
              f"wave-in-WG hist {np.bincount((bad//32)%4, minlength=4).tolist()} lane hist(first 8) {np.bincount(bad%32, minlength=32)[:8].tolist()} cols {cols} "
              f"maxerr {float(d.max()/sc):.2e} example rows {bad[:12].tolist()}", flush=True)
        r = int(bad[0]); print("   row", r, "got", cur[r].tolist(), "want", base[r].tolist(), flush=True)

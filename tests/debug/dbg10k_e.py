import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metatrain_amd import runtime as rt
from oracle import pet as opet
dev = torch.device("cuda:0")
hypers = dict(opet.DEFAULT_HYPERS)
params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")
n = 10000
pos, z, cell = opet.random_box(n, 0)
pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True]*3, 4.5)
graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), pairs[:,0].contiguous(), pairs[:,1].contiguous(), pairs[:,2:5].contiguous(), z.to(dev), torch.zeros(n, dtype=torch.int32, device=dev))
def run(f, b):
    fw = rt.HipForward(model, graph)
    rt.config_set("trr_compress", f); a = fw.forward(); torch.cuda.synchronize()
    rt.config_set("trr_compress", b); g = fw.backward(torch.ones_like(a)); torch.cuda.synchronize()
    return g
base = run(0, 0)
sc = float(base.abs().max())
for f, b in ((1, 0), (0, 1), (1, 1), (2, 0), (0, 2)):
    for rep in range(3):
        g = run(f, b)
        err = (g - base).abs().max(1).values / sc
        print(f"fwd={f} bwd={b} rep{rep}: max {float(err.max()):.2e} atoms>1e-5 {int((err > 1e-5).sum())}", flush=True)
rt.config_set("trr_compress", 3)

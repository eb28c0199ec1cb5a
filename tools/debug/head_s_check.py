import sys, torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params
dev = torch.device("cuda:0")
hypers = default_hypers()
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
for n in (700, 3000, 20000):
    pos, z, cell = random_box(n, 1)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5)
    S = torch.zeros(n, dtype=torch.int32, device=dev)
    graph = rt.HipGraph(model, pos.to(dev), cell.to(dev)[None], pairs[:, 0].contiguous(), pairs[:, 1].contiguous(), pairs[:, 2:5].contiguous(), z.to(dev), S)
    out = {}
    for mode in (0, 2 if n < 10000 else 1):
        rt.config_set("emlp_s", mode)
        fw = rt.HipForward(model, graph)
        a = fw.forward().clone(); g = fw.backward(torch.ones_like(a)).clone()
        out[mode] = (a, g)
    a0, g0 = out[0]; a2, g2 = out[2 if n < 10000 else 1]
    print(n, "E", float(a0.sum()), float(a2.sum()), "atomic rel", float((a0 - a2).abs().max() / a0.abs().max()), "grad rel", float((g0 - g2).abs().max() / g0.abs().max()))
rt.config_set("emlp_s", 1)

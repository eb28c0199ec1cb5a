import sys, torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt, data
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params
dev = torch.device("cuda:0")
hypers = default_hypers()
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
boxes = [random_box(1000, 100 + b) for b in range(64)]
def graph_of(bx):
    return data.graph_of(model, data.collate([(p.to(dev), z.to(dev), c.to(dev), (True, True, True)) for p, z, c in bx], 4.5))
g64, g16 = graph_of(boxes), graph_of(boxes[:16])
res = {}
for mode in (0, 1):
    rt.config_set("emlp_s", mode)
    for name, g in (("g64", g64), ("g16", g16)):
        for train in (False, True):
            fw = rt.HipForward(model, g, train=train)
            res[(mode, name, train)] = fw.forward().clone()
rt.config_set("emlp_s", 1)
ref = res[(0, "g64", False)]
for k, v in res.items():
    r = ref[: v.numel()]
    print(k, float((v - r).abs().max() / r.abs().max()))
import os
for rep in range(3):
    rt.config_set("emlp_s", 1)
    a_full = rt.HipForward(model, g64, train=True).forward().clone()
    a_full2 = rt.HipForward(model, g64, train=True).forward().clone()
    print("rep", rep, "full vs full", float((a_full - a_full2).abs().max()))
    for lo in range(0, 64, 16):
        g = graph_of(boxes[lo:lo + 16])
        a1 = rt.HipForward(model, g, train=True).forward().clone()
        a2 = rt.HipForward(model, g, train=True).forward().clone()
        r = a_full[lo * 1000:(lo + 16) * 1000]
        print("   lo", lo, "edges", g.n_edges, "rel", float((a1 - r).abs().max() / r.abs().max()), "run-to-run", float((a1 - a2).abs().max()))
rt.config_set("emlp_s", 1)

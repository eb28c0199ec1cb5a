"""Wall time of one graph build (and of one neighbour list) of an n-atom box, host call to host return, GPU otherwise idle:
what the read-backs inside them cost on the critical path of an MD step. python tools/debug/graph_build_time.py 1000 3000"""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params

dev = torch.device("cuda:0")
hypers = default_hypers()
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
for n in (int(a) for a in (sys.argv[1:] or ["1000"])):
    pos, z, cell = random_box(n, seed=0)
    posd, zd, celld = pos.to(dev), z.to(dev), cell.to(dev)[None]
    sysidx = torch.zeros(n, dtype=torch.int32, device=dev)
    pairs, _ = rt.neighbor_list(posd, cell, [True] * 3, hypers["cutoff"])
    i, j, s = pairs[:, 0].contiguous(), pairs[:, 1].contiguous(), pairs[:, 2:5].contiguous()
    out = {"atoms": n, "edges": int(pairs.shape[0])}
    for label, fn in (("graph_build_us", lambda: rt.HipGraph(model, posd, celld, i, j, s, zd, sysidx)),
                      ("neighbor_list_us", lambda: rt.neighbor_list(posd, cell, [True] * 3, hypers["cutoff"]))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        K = 300
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        out[label] = round((time.perf_counter() - t0) / K * 1e6, 1)
    print(json.dumps(out))

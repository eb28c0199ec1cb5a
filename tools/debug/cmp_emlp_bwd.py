"""dE/dR with the pipelined edge-MLP adjoint against the plain one (pet_config_set("emlp_bwd_pipe", 1 | 0))."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params
dev = torch.device("cuda:0")
hypers = default_hypers()
params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")
for n in [int(a) for a in sys.argv[1:]] or [64, 1000]:
    pos, z, cell = random_box(n, 3)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, hypers["cutoff"])
    graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), pairs[:, 0], pairs[:, 1], pairs[:, 2:5], z.to(dev),
                        torch.zeros(n, dtype=torch.int32, device=dev))
    out = {}
    for v in (0, 1):
        rt.config_set("emlp_bwd_pipe", v)
        fw = rt.HipForward(model, graph); a = fw.forward(); g = fw.backward(torch.ones_like(a)); torch.cuda.synchronize()
        out[v] = g.cpu().double().numpy()
    for v, name in ((1, "emlp_bwd_pipe"),):
        d = np.abs(out[v] - out[0]); s = np.abs(out[0]).max()
        bad = np.argwhere(d.max(1) > 1e-5 * s).ravel()
        print(f"n={n} edges={pairs.shape[0]} {name}: max|d|/max|g| = {d.max() / s:.3e}; atoms off: {len(bad)} first {bad[:12]}", flush=True)

"""A/B of the two forms of the fused attention adjoint (pet_config_set("attn_bwd_split", 1 | 0)) at 8 x 10k atoms and on a
small forced-fused box: largest difference of dE/dR between them, and the stage times."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params
dev = torch.device("cuda:0")
hypers = default_hypers()
params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")

def batch(nb, n):
    P, Z, C, PR, S = [], [], [], [], []
    for b in range(nb):
        pos, z, cell = random_box(n, b)
        pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5); pairs = pairs.clone(); pairs[:, :2] += b * n
        P.append(pos.to(dev)); Z.append(z.to(dev)); C.append(cell.to(dev)); PR.append(pairs)
        S.append(torch.full((n,), b, dtype=torch.int32, device=dev))
    P, Z, C, PR, S = torch.cat(P), torch.cat(Z), torch.stack(C), torch.cat(PR), torch.cat(S)
    return rt.HipGraph(model, P, C, PR[:, 0].contiguous(), PR[:, 1].contiguous(), PR[:, 2:5].contiguous(), Z, S), nb * n

rt.config_set("side_stream", 0)
for nb, n, force in ((1, 700, True), (8, 10000, False)):
    rt.config_set("attn_fused", 7 if force else 3)
    graph, na = batch(nb, n)
    fw = rt.HipForward(model, graph)
    ones = torch.ones(na, device=dev)
    res = {}
    for split in (0, 1):
        rt.config_set("attn_bwd_split", split)
        a = fw.forward(); g = fw.backward(ones).clone()
        torch.cuda.synchronize()
        g2 = fw.backward(ones)
        res[split] = (g, bool(torch.equal(g, g2)))
        rt.profile(True)
        for _ in range(5):
            fw.forward(); fw.backward(ones)
        torch.cuda.synchronize(); rep = rt.profile_report(); rt.profile(False)
        for r in rep:
            if r["name"].startswith("attn_blk"):
                print(f"{nb}x{n} split={split} {r['name']:14s} {r['total_ms'] / r['calls']:8.3f} ms per launch", flush=True)
    d = (res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max()
    print(f"{nb}x{n}: max |dE/dR(split) - dE/dR(one kernel)| / max |dE/dR| = {float(d):.2e}; finite {bool(torch.isfinite(res[1][0]).all())}; "
          f"run-to-run identical: one kernel {res[0][1]}, split {res[1][1]}", flush=True)
rt.config_set("attn_bwd_split", 1)

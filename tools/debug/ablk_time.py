"""Stage times of the fused attention kernels at 8 x 10k atoms (for ablation builds: no parity check)."""
import os, sys, time
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
nb = 8
P, Z, C, PR, S = [], [], [], [], []
for b in range(nb):
    pos, z, cell = random_box(10000, b)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5); pairs = pairs.clone(); pairs[:, :2] += b * 10000
    P.append(pos.to(dev)); Z.append(z.to(dev)); C.append(cell.to(dev)); PR.append(pairs)
    S.append(torch.full((10000,), b, dtype=torch.int32, device=dev))
P, Z, C, PR, S = torch.cat(P), torch.cat(Z), torch.stack(C), torch.cat(PR), torch.cat(S)
ones = torch.ones(nb * 10000, device=dev)
rt.config_set("attn_fused", int(os.environ.get("MODE", "3")))
rt.config_set("side_stream", 0)
graph = rt.HipGraph(model, P, C, PR[:, 0].contiguous(), PR[:, 1].contiguous(), PR[:, 2:5].contiguous(), Z, S)
fw = rt.HipForward(model, graph)
for _ in range(3):
    a = fw.forward(); g = fw.backward(ones)
torch.cuda.synchronize()
rt.profile(True)
for _ in range(5):
    a = fw.forward(); g = fw.backward(ones)
torch.cuda.synchronize(); rep = rt.profile_report(); rt.profile(False)
for r in rep:
    if r["name"].startswith("attn_blk") or r["name"] in ("qkv", "attn_fwd", "oproj", "emlp"):
        print(f"{os.environ.get('TAG', '')} {r['name']:14s} {r['total_ms'] / r['calls']:8.3f} ms per launch", flush=True)

"""8 x 10k atoms, forward + adjoint a few times with the split adjoint (for rocprofv3 passes)."""
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
nb, n = 8, 10000
P, Z, C, PR, S = [], [], [], [], []
for b in range(nb):
    pos, z, cell = random_box(n, b)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5); pairs = pairs.clone(); pairs[:, :2] += b * n
    P.append(pos.to(dev)); Z.append(z.to(dev)); C.append(cell.to(dev)); PR.append(pairs)
    S.append(torch.full((n,), b, dtype=torch.int32, device=dev))
P, Z, C, PR, S = torch.cat(P), torch.cat(Z), torch.stack(C), torch.cat(PR), torch.cat(S)
graph = rt.HipGraph(model, P, C, PR[:, 0].contiguous(), PR[:, 1].contiguous(), PR[:, 2:5].contiguous(), Z, S)
rt.config_set("side_stream", 0)
rt.config_set("attn_bwd_split", int(os.environ.get("SPLIT", "1")))
fw = rt.HipForward(model, graph)
ones = torch.ones(nb * n, device=dev)
for _ in range(int(os.environ.get("REPS", "3"))):
    fw.forward(); fw.backward(ones)
torch.cuda.synchronize()

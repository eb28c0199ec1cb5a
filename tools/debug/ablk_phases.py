"""Per-phase cycle sums of the fused attention kernels (library built with PET_HIP_EXTRA_FLAGS=-DAB_PROFILE):
one forward + adjoint of 8 x 10k atoms, then the dump (stderr)."""
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
nb = 8
P, Z, C, PR, S = [], [], [], [], []
for b in range(nb):
    pos, z, cell = random_box(10000, b)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, 4.5); pairs = pairs.clone(); pairs[:, :2] += b * 10000
    P.append(pos.to(dev)); Z.append(z.to(dev)); C.append(cell.to(dev)); PR.append(pairs)
    S.append(torch.full((10000,), b, dtype=torch.int32, device=dev))
P, Z, C, PR, S = torch.cat(P), torch.cat(Z), torch.stack(C), torch.cat(PR), torch.cat(S)
rt.config_set("attn_fused", 3)
graph = rt.HipGraph(model, P, C, PR[:, 0].contiguous(), PR[:, 1].contiguous(), PR[:, 2:5].contiguous(), Z, S)
fw = rt.HipForward(model, graph)
for it in range(2):
    a = fw.forward(); torch.cuda.synchronize()
    if it == 1: print("forward phases (4 launches):", file=sys.stderr, flush=True)
    rt.config_set("attn_fused_prof", 0)
    g = fw.backward(torch.ones_like(a)); torch.cuda.synchronize()
    if it == 1: print("adjoint phases (4 launches):", file=sys.stderr, flush=True)
    rt.config_set("attn_fused_prof", 0)

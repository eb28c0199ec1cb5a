"""One-off check of 64-bit indexing: a 400 000-atom box (7.6 M edges: activation offsets beyond 2^32 elements) evaluated
whole on one GPU against the sum of its 8-rank partition (sub-systems of 104 k atoms, offsets below 2^31)."""
import sys
import torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers, partition
from metatrain_amd.synthetic import random_box, synthetic_params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
dev = torch.device("cuda:0")
hypers = default_hypers()
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
pos, z, cell = random_box(n, seed=0)
posd, zd = pos.to(dev), z.to(dev)
e_ref, g_ref, _, _ = partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, 1, 0)
e_ref, g_ref = float(e_ref), g_ref.clone()
torch.cuda.empty_cache()
e, grad = 0.0, torch.zeros_like(g_ref)
for r in range(8):
    er, gr, n_sub, n_owned = partition.energy_and_gradient(model, posd, zd, cell, [True] * 3, 8, r)
    e += float(er); grad += gr
print("atoms", n, "E whole", e_ref, "E parts", e, "rel", abs(e - e_ref) / abs(e_ref))
print("grad max rel diff", float((grad - g_ref).abs().max() / g_ref.abs().max()))

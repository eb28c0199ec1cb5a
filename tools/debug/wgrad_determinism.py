"""Run the second-order pass twice and report where the flat gradient differs (debug aid)."""
import sys
import torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params

dev = torch.device("cuda:0")
hypers = default_hypers()
params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in params.items()}, "energy")
n = 1000
pos, z, cell = random_box(n, seed=9)
pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, hypers["cutoff"])
graph = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                    pairs[:, 2:5].contiguous(), z.to(dev), torch.zeros(n, dtype=torch.int32, device=dev))
fw = rt.HipForward(model, graph, train=True)
fw.forward()
ones = torch.ones(n, device=dev)
gen = torch.Generator().manual_seed(0)
u = torch.randn(n, 3, generator=gen).to(dev)
nu = torch.rand(n, generator=gen).to(dev)
for mode, side in ((1, 1),):
    rt.config_set("wgrad_bf16", mode)
    rt.config_set("side_stream", side)
    print("== wgrad_bf16", mode, "side_stream", side)
    runs = []
    for rep in range(3):
        model.zero_grad()
        fw.backward_train(ones)
        runs.append({k: v.clone() for k, v in model.grads().items()})
    for k in runs[0]:
        for rep in (1, 2):
            if not torch.equal(runs[0][k], runs[rep][k]):
                d = (runs[0][k] != runs[rep][k])
                print("wgrad_bf16", mode, k, tuple(runs[0][k].shape), "rep", rep, "differing", int(d.sum()),
                      "max abs diff", float((runs[0][k] - runs[rep][k]).abs().max()), "max", float(runs[0][k].abs().max()))
                if d.dim() == 2 and rep == 1:
                    print("    per-column counts (nonzero only):", {int(c): int(d[:, c].sum()) for c in d.any(0).nonzero().flatten()[:12]})
                    print("    per-row counts:", sorted(set(d.sum(1).tolist()))[:10])
                    print("    cols differing:", d.any(0).nonzero().flatten().tolist())
                if d.dim() == 2:
                    rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
                    print("    rows", rows[:6].tolist(), "...", rows[-3:].tolist(), "cols", cols[:6].tolist(), "...", cols[-3:].tolist())
print("done")

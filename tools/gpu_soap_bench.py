"""Time the SOAP-BPNN path (forward + dE/dR) on one n-atom random box; per-stage table. GPU only."""
import sys
import time

import torch

sys.path.insert(0, ".")
from metatrain_amd import runtime as rt  # noqa: E402
from metatrain_amd.soap_bpnn import SoapBpnnHip, default_hypers  # noqa: E402
from metatrain_amd.synthetic import random_box  # noqa: E402

natoms = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
legacy = (sys.argv[2] != "alchemical") if len(sys.argv) > 2 else True
dev = torch.device("cuda:0")
hypers = default_hypers()
hypers["legacy"] = legacy
model = SoapBpnnHip(hypers, [1, 6, 7, 8])
S, H = model.feature_size, 32
gen = torch.Generator().manual_seed(0)
params = {}
n_sets = 4 if legacy else 1
if not legacy:
    params["species_embedding.weight"] = torch.randn(4, 4, generator=gen)
    params["center_encoding.weight"] = torch.randn(4, S, generator=gen)
for s in range(n_sets):
    params[f"layernorm.{s}.weight"] = 1 + 0.1 * torch.randn(S, generator=gen)
    params[f"layernorm.{s}.bias"] = 0.1 * torch.randn(S, generator=gen)
    params[f"bpnn.{s}.0.weight"] = torch.randn(H, S, generator=gen) / S**0.5
    params[f"bpnn.{s}.2.weight"] = torch.randn(H, H, generator=gen) / H**0.5
    params[f"last_layers.energy.{s}.weight"] = torch.randn(1, H, generator=gen) / H**0.5
model.load({k: v.to(dev) for k, v in params.items()})
pos, z, cell = random_box(natoms, seed=0)
posd = pos.to(dev)
pairs, _ = rt.neighbor_list(posd, cell, [True] * 3, 5.0)
sysidx = torch.zeros(natoms, dtype=torch.int32, device=dev)
args = (posd, cell[None].to(dev), pairs[:, 0].contiguous(), pairs[:, 1].contiguous(), pairs[:, 2:5].contiguous(),
        z.to(dev), sysidx)
ones = torch.ones(natoms, device=dev)


def step():
    g = model.graph(*args)
    a = model.forward(g)
    return a, model.backward(g, ones), g


a, grad, g = step()
torch.cuda.synchronize()
print("atoms", natoms, "pairs", g.n_edges, "E", float(a.double().sum()), "max|F|", float(grad.abs().max()))
t0 = time.perf_counter()
n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("step %.3f ms  -> %.0f atom-steps/s" % (dt * 1e3, natoms / dt))
rt.profile(True)
step()
torch.cuda.synchronize()
for r in sorted(rt.profile_report(), key=lambda r: -r["total_ms"]):
    print("  %-18s %8.3f ms  %8.1f GFLOP/s  %8.1f GB/s(alg)" % (r["name"], r["total_ms"], r["flops"] / max(r["total_ms"], 1e-9) / 1e6,
                                                                 r["bytes"] / max(r["total_ms"], 1e-9) / 1e6))

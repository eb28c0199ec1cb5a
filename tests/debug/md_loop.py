"""One-off: an MD-like loop on one box (10 000 atoms, or argv[2]) -- positions move a little every step, the neighbour list,
graph and edge count change, one activation workspace is reused (HipForward.rebind) -- against a fresh workspace every 10
steps; memory in use must not grow. Prints a checksum of every step's energies and forces: two runs of the same command must
print the same one (the graph build's polled read-back and its assumed sort order, round 6, are on this path every step).
python tests/debug/md_loop.py 1000 1000"""
import sys
import time
import torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params

dev = torch.device("cuda:0")
hypers = default_hypers()
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
NAT = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
pos, z, cell = random_box(NAT, seed=0)
pos, z = pos.to(dev), z.to(dev)
sysidx = torch.zeros(NAT, dtype=torch.int32, device=dev)
gen = torch.Generator(device=dev).manual_seed(0)


def graph_of(p):
    pairs, _ = rt.neighbor_list(p, cell, [True] * 3, hypers["cutoff"])
    return rt.HipGraph(model, p, cell[None].to(dev), pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                       pairs[:, 2:5].contiguous(), z, sysidx)


g0 = graph_of(pos)
fw = rt.HipForward(model, g0)
cap_edges = int(g0.n_edges * 1.05)
ones = torch.ones(NAT, device=dev)
chk = torch.zeros(2, dtype=torch.float64, device=dev)
mem0 = None
t0 = time.perf_counter()
worst = 0.0
edges = set()
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    pos = pos + 0.02 * torch.randn(pos.shape, device=dev, generator=gen)
    g = graph_of(pos)
    edges.add(g.n_edges)
    try:
        fw.rebind(g)
    except rt.PetHipError:
        fw = rt.HipForward(model, g)
    a = fw.forward()
    f = fw.backward(ones)
    if step % 10 == 0:
        fw2 = rt.HipForward(model, g)
        a2 = fw2.forward(); f2 = fw2.backward(ones)
        assert torch.equal(a, a2) and torch.equal(f, f2), "reused workspace differs from a fresh one"
        del fw2
        torch.cuda.synchronize()
        mem = torch.cuda.memory_allocated()
        mem0 = mem0 or mem
        worst = max(worst, mem / mem0)
    assert torch.isfinite(f).all()
    chk += torch.stack([a.double().sum(), (f.double() * f.double()).sum()])
torch.cuda.synchronize()
print("steps ok; distinct edge counts", len(edges), "ms/step", (time.perf_counter() - t0) / (step + 1) * 1e3,
      "memory growth x", round(worst, 3), "checksum", [float(x).hex() for x in chk.cpu()])

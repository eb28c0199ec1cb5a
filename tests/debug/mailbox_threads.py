"""One-off: graph builds and neighbour lists from several host threads at once (each thread polls its own pinned mailbox for the
counts its launches publish: csrc/graph.hip read_back); every thread's results must equal the single-threaded ones."""
import sys, threading
sys.path.insert(0, ".")
import torch
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params

dev = torch.device("cuda:0")
hypers = default_hypers()
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
boxes = []
for seed, n in ((1, 500), (2, 1200), (3, 3000), (4, 800)):
    pos, z, cell = random_box(n, seed=seed)
    boxes.append((pos.to(dev), z.to(dev), cell, torch.zeros(n, dtype=torch.int32, device=dev)))


def one(b):
    pos, z, cell, sysidx = b
    pairs, _ = rt.neighbor_list(pos, cell, [True] * 3, hypers["cutoff"])
    g = rt.HipGraph(model, pos, cell[None].to(dev), pairs[:, 0].contiguous(), pairs[:, 1].contiguous(), pairs[:, 2:5].contiguous(),
                    z, sysidx)
    return int(pairs.shape[0]), int(g.n_edges), int(g.csr()["rowptr"].sum())


ref = [one(b) for b in boxes]
bad = []


def worker(k):
    for it in range(150):
        b = (k + it) % len(boxes)
        if one(boxes[b]) != ref[b]:
            bad.append((k, it))


ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
[t.start() for t in ts]
[t.join() for t in ts]
print("reference", ref, "mismatches", bad)
assert not bad

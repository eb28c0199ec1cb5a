"""Time the pieces of the training step (energy-loss term) on B x 10k-atom boxes. GPU only."""
import sys
import time

import torch

sys.path.insert(0, ".")
from metatrain_amd import runtime as rt  # noqa: E402
from metatrain_amd.pet import default_hypers  # noqa: E402
from metatrain_amd.synthetic import random_box, synthetic_params  # noqa: E402

boxes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
natoms = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda:0")
hypers = default_hypers()
params = synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0)
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in params.items()}, "energy")
pos_l, z_l, cell_l, pair_l, sys_l = [], [], [], [], []
for b in range(boxes):
    pos, z, cell = random_box(natoms, seed=b)
    posd = pos.to(dev)
    pairs, _ = rt.neighbor_list(posd, cell, [True] * 3, hypers["cutoff"])
    pairs = pairs.clone()
    pairs[:, 0:2] += b * natoms
    pos_l.append(posd); z_l.append(z.to(dev)); cell_l.append(cell.to(dev)); pair_l.append(pairs)
    sys_l.append(torch.full((natoms,), b, dtype=torch.int32, device=dev))
positions, species, cells = torch.cat(pos_l), torch.cat(z_l), torch.stack(cell_l)
pairs, sysidx = torch.cat(pair_l), torch.cat(sys_l)
graph = rt.HipGraph(model, positions, cells, pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                    pairs[:, 2:5].contiguous(), species, sysidx)
print("atoms", boxes * natoms, "edges", graph.n_edges)
fw = rt.HipForward(model, graph, train=True)
print("workspace GB", fw.nbytes / 1e9)
seeds = torch.ones(boxes * natoms, device=dev)
model.zero_grad()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("forward            %.2f ms" % timeit(lambda: fw.forward()))
print("backward (forces)  %.2f ms" % timeit(lambda: fw.backward(seeds)))
print("backward_train     %.2f ms" % timeit(lambda: fw.backward_train(seeds, want_position_grad=True)))
step = [0]


def adam():
    step[0] += 1
    model.adam_step(1e-4, step[0], max_grad_norm=1.0)


u = torch.randn(boxes * natoms, 3, device=dev) * 1e-3
print("backward_train2    %.2f ms" % timeit(lambda: fw.backward_train2(seeds, seeds, u), 3))
print("workspace2 GB", fw.workspace2.numel() / 1e9)
print("clip+adam+repack   %.2f ms" % timeit(adam))
rt.profile(True)
fw.forward(); fw.backward(seeds); fw.backward_train2(seeds, seeds, u)
torch.cuda.synchronize()
for r in sorted(rt.profile_report(), key=lambda r: -r["total_ms"])[:16]:
    print("  %-16s %8.3f ms x%-3d %8.1f TF/s" % (r["name"], r["total_ms"], r["calls"], r["flops"] / max(r["total_ms"], 1e-9) / 1e9))

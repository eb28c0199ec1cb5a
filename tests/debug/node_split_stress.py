"""Stress of the four-workgroups-per-row-tile node kernels (k_node2 / k_node_bwd2, SPLIT): the partial tiles travel between
workgroups on different XCDs through device-coherent stores and loads, without a cache write-back fence. A stale read would
show as a run that differs from the first: energies and dE/dR of one graph, N_REP times, must be bit-identical, for several
box sizes (every tile count of the split up to its 128-tile limit).   python tests/debug/node_split_stress.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from metatrain_amd import runtime as rt  # noqa: E402
from metatrain_amd.pet import default_hypers  # noqa: E402
from metatrain_amd.synthetic import random_box, synthetic_params  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
hypers = default_hypers()
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
bad = 0
for n in (33, 1000, 2500, 4096):
    pos, z, cell = random_box(n, seed=n)
    posd = pos.to(dev)
    pairs, _ = rt.neighbor_list(posd, cell, [True] * 3, hypers["cutoff"])
    sysidx = torch.zeros(n, dtype=torch.int32, device=dev)
    g = rt.HipGraph(model, posd, cell.to(dev)[None], pairs[:, 0].contiguous(), pairs[:, 1].contiguous(),
                    pairs[:, 2:5].contiguous(), z.to(dev), sysidx)
    fw = rt.HipForward(model, g)
    ones = torch.ones(n, device=dev)
    a0 = fw.forward().clone()
    g0 = fw.backward(ones).clone()
    rt.config_set("node_split", 0)
    a1 = fw.forward().clone()
    g1 = fw.backward(ones).clone()
    rt.config_set("node_split", 1)
    fwd_same = bool(torch.equal(a0, a1))
    rel = float((g0 - g1).abs().max() / g1.abs().max())
    mism = 0
    for _ in range(reps):
        a = fw.forward()
        gr = fw.backward(ones)
        if not (torch.equal(a, a0) and torch.equal(gr, g0)):
            mism += 1
    bad += mism + (0 if fwd_same else 1)
    print(f"{n:5d} atoms: {mism} of {reps} runs differ from the first; forward bit-identical to the unsplit kernel: {fwd_same}; "
          f"dE/dR split vs unsplit {rel:.1e}")
sys.exit(1 if bad else 0)

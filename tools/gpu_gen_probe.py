"""Size-generic path (csrc/gen.hip) at sizes other than the tuned one: forward + dE/dR of one N-atom box, wall time and the
per-kernel share (rt.profile stages if instrumented; otherwise wall only).  python tools/gpu_gen_probe.py [N]"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params

SIZES = {"s64": dict(d_pet=64, d_node=128, d_feedforward=128, d_head=64, num_heads=4),
         "wide256": dict(d_pet=256, d_node=512, d_feedforward=320, d_head=96, num_heads=4),
         "default": {}}
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for tag, delta in SIZES.items():
    hypers = dict(default_hypers(), **delta)
    model = rt.HipModel(hypers, [1, 6, 7, 8])
    model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
    pos, z, cell = random_box(n, seed=0)
    pr, _ = rt.neighbor_list(pos.to(dev), cell, [True] * 3, hypers["cutoff"])
    g = rt.HipGraph(model, pos.to(dev), cell[None].to(dev), pr[:, 0].contiguous(), pr[:, 1].contiguous(), pr[:, 2:5].contiguous(),
                    z.to(dev), torch.zeros(n, dtype=torch.int32, device=dev))
    fw = rt.HipForward(model, g)
    ones = torch.ones(n, device=dev)
    for _ in range(2):
        fw.forward(); fw.backward(ones)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        fw.forward(); fw.backward(ones)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(json.dumps({"size": tag, "atoms": n, "edges": int(g.n_edges), "fwd+bwd_ms": dt * 1e3, "atom_steps_per_s": n / dt}), flush=True)

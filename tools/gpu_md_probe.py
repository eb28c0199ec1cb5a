"""MD regime: ONE box per step (neighbour list + graph + forward + dE/dR), wall time per step against the sum of the
kernel times (single stream, profile events): how much of a small box's step is launch / host overhead."""
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import torch
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params

dev = torch.device("cuda:0")
hypers = default_hypers()
for kv in os.environ.get("SET", "").split(","):   # library switches for A/B runs: SET=emlp_s=0,attn_fused=0
    if kv:
        rt.config_set(kv.split("=")[0], int(kv.split("=")[1]))
model = rt.HipModel(hypers, [1, 6, 7, 8])
model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
for n in (int(a) for a in (sys.argv[1:] or ["1000", "10000"])):
    pos, z, cell = random_box(n, seed=0)
    posd, zd, celld = pos.to(dev), z.to(dev), cell.to(dev)[None]
    sysidx = torch.zeros(n, dtype=torch.int32, device=dev)
    ones = torch.ones(n, device=dev)
    state = {}

    def step(with_nl=True):
        if with_nl or "pairs" not in state:
            state["pairs"], _ = rt.neighbor_list(posd, cell, [True] * 3, hypers["cutoff"])
        p = state["pairs"]
        g = rt.HipGraph(model, posd, celld, p[:, 0].contiguous(), p[:, 1].contiguous(), p[:, 2:5].contiguous(), zd, sysidx)
        fw = state.get("fw")
        fw = fw.rebind(g) if fw is not None else rt.HipForward(model, g)
        state["fw"] = fw
        a = fw.forward()
        return a, fw.backward(ones)

    out = {}
    for label, kw in (("nl+graph+fwd+bwd", dict(with_nl=True)), ("graph+fwd+bwd", dict(with_nl=False))):
        for _ in range(5):
            step(**kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 50
        for _ in range(K):
            step(**kw)
        torch.cuda.synchronize()
        out[label + "_ms"] = (time.perf_counter() - t0) / K * 1e3
    rt.config_set("side_stream", 0)
    rt.profile(True)
    step(with_nl=False)
    torch.cuda.synchronize()
    out["stage_kernels_ms"] = sum(r["total_ms"] for r in rt.profile_report())
    rt.profile(False)
    rt.config_set("side_stream", 1)
    out["atoms"] = n
    out["atom_steps_per_s"] = n / out["nl+graph+fwd+bwd_ms"] * 1e3
    print(json.dumps(out))

"""Run-to-run bit determinism of forward (+ adjoint) on batches that hand every stage to the large-graph kernels:
64 x 1 000 atoms (training and inference forward) and 4 x 10 000 atoms, N runs each; prints how many runs differ from the first."""
import sys, torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt, data
from metatrain_amd.pet import default_hypers
from metatrain_amd.synthetic import random_box, synthetic_params
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
hypers = default_hypers()
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0).items()}, "energy")
def graph_of(n_box, n_at):
    bx = [random_box(n_at, 100 + b) for b in range(n_box)]
    return data.graph_of(model, data.collate([(p.to(dev), z.to(dev), c.to(dev), (True, True, True)) for p, z, c in bx], 4.5))
for name, g in (("64x1000", graph_of(64, 1000)), ("4x10000", graph_of(4, 10000))):
    for train in (False, True):
        for side in (1, 0):
            rt.config_set("side_stream", side)
            fw = rt.HipForward(model, g, train=train)
            ref = fw.forward().clone()
            gref = fw.backward(torch.ones_like(ref)).clone()
            bad_f = bad_b = 0
            for it in range(N):
                a = fw.forward()
                bad_f += int(not torch.equal(a, ref))
                gr = fw.backward(torch.ones_like(a))
                bad_b += int(not torch.equal(gr, gref))
            print(name, "train", train, "side_stream", side, f"forward runs differing: {bad_f} of {N}; adjoint runs differing: {bad_b} of {N}", flush=True)
rt.config_set("side_stream", 1)

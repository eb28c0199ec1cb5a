import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T
from oracle import pet as opet
from metatrain_amd import runtime as rt
from metatrain_amd.pet.trainer import TrainStep
dev = torch.device("cuda:0")
hypers = dict(opet.DEFAULT_HYPERS); types = [1, 6, 7, 8]
params = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float32)
inp = T._inputs(os.path.join(ROOT, "tests", "golden"), "batch_two_systems.npz")
s = inp["system_indices"].long()
n_atoms = torch.bincount(s).float().to(dev)
targets = (torch.tensor([1.5, -2.0]) * n_atoms.cpu()).to(dev)
tg = (0.3 * torch.randn(len(s), 3, generator=torch.Generator().manual_seed(3))).to(dev)
th = {"learning_rate": 1e-3, "warmup_fraction": 0.5, "num_epochs": 6}
def fresh(weights):
    model = rt.HipModel(hypers, types)
    model.load({k: v.to(dev) for k, v in weights.items()}, "energy")
    graph = rt.HipGraph(model, inp["positions"].float().to(dev), inp["cells"].float().to(dev), inp["centers"].to(dev),
                        inp["neighbors"].to(dev), inp["cell_shifts"].to(dev), inp["species"].to(dev), inp["system_indices"].int().to(dev))
    return model, graph, rt.HipForward(model, graph, train=True), TrainStep(model, th)
def run():
    model, graph, fw, step = fresh(params)
    for _ in range(2):
        step(graph, fw, targets, n_atoms, tg)
    ckpt = {"trainer": step.state_dict(), "weights": {k: v.cpu() for k, v in model.state_dict().items()}}
    out_a = step(graph, fw, targets, n_atoms, tg)
    final_a = model.state_dict()
    weights = dict(params); weights.update(ckpt["weights"])
    model_b, graph_b, fw_b, step_b = fresh(weights)
    step_b.load_state_dict(ckpt["trainer"])
    out_b = step_b(graph_b, fw_b, targets, n_atoms, tg)
    final_b = model_b.state_dict()
    bad = [k for k in final_a if not torch.equal(final_a[k], final_b[k])]
    return float(out_a["loss"]) == float(out_b["loss"]), bad
for mode in (3, 0, 3):
    rt.config_set("attn_fused", mode)
    res = [run() for _ in range(8)]
    print("attn_fused", mode, "fails", sum(1 for ok, bad in res if bad), "of 8;", [len(b) for _, b in res], [b[:2] for _, b in res if b][:2])
rt.config_set("side_stream", 0)
res = [run() for _ in range(8)]
print("side_stream 0: fails", sum(1 for ok, bad in res if bad), "of 8")

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
# same model, same step repeated from identical state: are the gradients reproducible at all?
model, graph, fw, step = fresh(params)
gs = []
for rep in range(4):
    model.zero_grad()
    a = fw.forward(); e = fw.sum_over_atoms(a); gp = fw.backward(torch.ones_like(a))
    fw.backward_train2(torch.ones_like(a), torch.ones_like(a) * 0.1, tg)
    gs.append(model.flat_grad().clone())
for rep in range(1, 4):
    d = (gs[rep] - gs[0]).abs()
    print("repeat", rep, "max diff", float(d.max()), "n diff", int((d > 0).sum()))
m2, g2, fw2, _ = fresh(params)
m2.zero_grad(); a = fw2.forward(); fw2.backward(torch.ones_like(a)); fw2.backward_train2(torch.ones_like(a), torch.ones_like(a) * 0.1, tg)
d = (m2.flat_grad() - gs[0]).abs(); print("fresh model", float(d.max()), int((d > 0).sum()))
offs = model.grad_offsets() if hasattr(model, "grad_offsets") else None
if int((d > 0).sum()) and offs is None:
    idx = torch.nonzero(d > 0).reshape(-1)[:10]; print("first differing flat indices", idx.tolist())
    gr_a = {k: v.clone() for k, v in model.grads().items()}
    gr_b = m2.grads()
    for k in gr_a:
        dd = (gr_a[k] - gr_b[k]).abs()
        if float(dd.max()) > 0: print("  differs:", k, float(dd.max()), int((dd > 0).sum()), "of", dd.numel())

print("---- resume flow")
model, graph, fw, step = fresh(params)
for _ in range(2):
    step(graph, fw, targets, n_atoms, tg)
ckpt = {"trainer": step.state_dict(), "weights": {k: v.cpu() for k, v in model.state_dict().items()}}
weights = dict(params); weights.update(ckpt["weights"])
model_b, graph_b, fw_b, step_b = fresh(weights)
step_b.load_state_dict(ckpt["trainer"])
sa, sb = model.state_dict(), model_b.state_dict()
print("weights equal before step 3:", all(torch.equal(sa[k], sb[k]) for k in sa))
oa, ob = model.optimizer_state(), model_b.optimizer_state()
for k in oa:
    if torch.is_tensor(oa[k]): print("optimizer", k, "equal:", torch.equal(oa[k], ob[k]))
    else: print("optimizer", k, oa[k], ob[k])
def grads_of(m, g, f):
    m.zero_grad(); a = f.forward(); f.backward(torch.ones_like(a)); f.backward_train2(torch.ones_like(a), torch.ones_like(a) * 0.1, tg)
    return m.flat_grad().clone(), a.clone()
ga, aa = grads_of(model, graph, fw); gb, ab = grads_of(model_b, graph_b, fw_b)
print("atomic equal:", torch.equal(aa, ab), "grad diff", float((ga - gb).abs().max()), int(((ga - gb).abs() > 0).sum()))
rt.config_set("attn_fused", 0)
ga, aa = grads_of(model, graph, fw); gb, ab = grads_of(model_b, graph_b, fw_b)
print("attn_fused=0: atomic equal:", torch.equal(aa, ab), "grad diff", float((ga - gb).abs().max()), int(((ga - gb).abs() > 0).sum()))

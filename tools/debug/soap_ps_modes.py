import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import nl as onl, soap as osoap
from metatrain_amd import runtime as rt
from metatrain_amd.soap_bpnn import SoapBpnnHip
dev = torch.device("cuda:0")
def box(n, seed):
    gen = torch.Generator().manual_seed(seed)
    L = (n / 0.05) ** (1 / 3)
    pos = torch.rand(n, 3, generator=gen, dtype=torch.float64) * L
    z = torch.tensor([1, 6, 7, 8])[torch.randint(0, 4, (n,), generator=gen)]
    cell = torch.eye(3, dtype=torch.float64) * L
    i, j, s, _ = onl.neighbor_list(pos.numpy(), cell.numpy(), [True] * 3, 5.0)
    return pos, z, cell[None], torch.tensor(i), torch.tensor(j), torch.tensor(s).long(), torch.zeros(n, dtype=torch.long)
rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(np.asarray(b, np.float64)).max())
for n in (96, 150):
  for legacy in (True, False):
    hypers = dict(osoap.DEFAULT_HYPERS, legacy=legacy); types = [1, 6, 7, 8]
    params = osoap.synthetic_params(hypers, 4, osoap.basis(hypers)[0], 0, torch.float32)
    pos, z, cells, ci, cj, cs, sysidx = box(n, 9)
    p64 = {k: v.double() for k, v in params.items()}
    _, g_ref, a_ref = osoap.energy_and_gradient(p64, hypers, types, pos, cells, ci, cj, cs, z, sysidx)
    model = SoapBpnnHip(hypers, types); model.load({k: v.to(dev) for k, v in params.items()})
    g = model.graph(pos.float().to(dev), cells.float().to(dev), ci.to(dev), cj.to(dev), cs.to(dev), z.to(dev), sysidx.int().to(dev))
    for order in ((1, 0), (0, 1)):
        for mode in order:
            rt.config_set("soap_ps_mfma", mode)
            atomic, feats = model.forward(g, want_features=True)
            grad = model.backward(g, torch.ones_like(atomic))
            print(n, legacy, "mode", mode, "E", rel(atomic.cpu(), a_ref), "grad", rel(grad.cpu(), g_ref), flush=True)
    rt.config_set("soap_ps_mfma", 1)

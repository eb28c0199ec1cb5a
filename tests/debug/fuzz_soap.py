"""One-off randomized parity sweep of SOAP-BPNN (energy + dE/dR) against the fp64 oracle: random sizes, densities,
triclinic cells, mixed periodicity, both species treatments."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from metatrain_amd.soap_bpnn import SoapBpnnHip
from oracle import nl as onl
from oracle import soap as osoap

dev = torch.device("cuda:0")
types = [1, 6, 7, 8]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = (0.0, 0.0)
models = {}
for legacy in (True, False):
    hypers = dict(osoap.DEFAULT_HYPERS, legacy=legacy)
    params = osoap.synthetic_params(hypers, 4, osoap.basis(hypers)[0], 0, torch.float32)
    m = SoapBpnnHip(hypers, types)
    m.load({k: v.to(dev) for k, v in params.items()})
    models[legacy] = (hypers, params, m)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    legacy = bool(rng.random() < 0.5)
    hypers, params, model = models[legacy]
    n = int(rng.integers(2, 150))
    rho = float(10 ** rng.uniform(-2.4, -1.0))
    L = max((n / rho) ** (1 / 3), 3.5)
    cell = np.eye(3) * L + (rng.uniform(-0.25, 0.25, (3, 3)) * L if rng.random() < 0.5 else 0.0)
    pbc = [bool(b) for b in rng.random(3) < 0.7]
    pos = torch.tensor(rng.random((n, 3)) @ cell, dtype=torch.float32)
    cells = torch.tensor(cell, dtype=torch.float32)[None]
    z = torch.tensor(rng.choice(types, n))
    i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cells[0].double().numpy(), pbc, 5.0)
    if len(i) == 0 or np.bincount(i, minlength=n).max() > 150:
        continue
    ci, cj, cs = torch.tensor(i, dtype=torch.int64), torch.tensor(j, dtype=torch.int64), torch.tensor(s, dtype=torch.int64).reshape(-1, 3)
    sysidx = torch.zeros(n, dtype=torch.int64)
    p64 = {k: v.double() for k, v in params.items()}
    e_ref, g_ref, a_ref = osoap.energy_and_gradient(p64, hypers, types, pos.double(), cells.double(), ci, cj, cs, z, sysidx)
    g = model.graph(pos.to(dev), cells.to(dev), ci.to(dev), cj.to(dev), cs.to(dev), z.to(dev), sysidx.int().to(dev))
    atomic = model.forward(g)
    grad = model.backward(g, torch.ones_like(atomic))
    ea = float((atomic.cpu().double() - a_ref).abs().max() / a_ref.abs().max())
    eg = float((grad.cpu().double() - g_ref).abs().max() / g_ref.abs().max())
    worst = (max(worst[0], ea), max(worst[1], eg))
    flag = "" if ea < 1e-5 and eg < 1e-5 else "   <-- ABOVE 1e-5"
    if flag:  # yardstick: the oracle itself in fp32
        e32, g32, a32 = osoap.energy_and_gradient(params, hypers, types, pos, cells, ci, cj, cs, z, sysidx)
        flag += f" (oracle fp32: E {float((a32.double() - a_ref).abs().max() / a_ref.abs().max()):.2e} grad " \
                f"{float((g32.double() - g_ref).abs().max() / g_ref.abs().max()):.2e})"
    print(f"trial {trial:3d} legacy={legacy} n={n} rho={rho:.4f} pbc={pbc} pairs {len(i)}: E {ea:.2e} grad {eg:.2e}{flag}", flush=True)
print("worst", worst)

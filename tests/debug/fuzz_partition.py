"""One-off randomized sweep of the one-box partitions (PET and SOAP-BPNN): random cells, periodicity and rank counts;
the partial results summed over the ranks against the whole box (same kernels, so this checks the partition logic)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt
from metatrain_amd.pet import default_hypers, partition as ppart
from metatrain_amd.soap_bpnn import SoapBpnnHip, partition as spart
from metatrain_amd.synthetic import synthetic_params
from oracle import soap as osoap

dev = torch.device("cuda:0")
types = [1, 6, 7, 8]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
pets = {}
for adaptive in (False, True):
    h = default_hypers()
    if adaptive:
        h.update(num_neighbors_adaptive=12, adaptive_cutoff_method="solver", cutoff_width_adaptive=1.0)
    m = rt.HipModel(h, types)
    m.load({k: v.to(dev) for k, v in synthetic_params(h, types, {"energy": 1}, 0).items()}, "energy")
    pets[adaptive] = m
sh = dict(osoap.DEFAULT_HYPERS)
soap = SoapBpnnHip(sh, types)
soap.load({k: v.to(dev) for k, v in osoap.synthetic_params(sh, 4, osoap.basis(sh)[0], 0, torch.float32).items()})
worst = 0.0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    n = int(rng.integers(2000, 12000))
    L = rng.uniform(0.6, 1.8, 3)
    L *= (n / 0.05 / L.prod()) ** (1 / 3)
    cell = np.diag(L) + (rng.uniform(-0.2, 0.2, (3, 3)) * L.min() if rng.random() < 0.5 else 0.0)
    pbc = [bool(b) for b in rng.random(3) < 0.75]
    pos = torch.tensor(rng.random((n, 3)) @ cell, dtype=torch.float32, device=dev)
    z = torch.tensor(rng.choice(types, n), device=dev)
    c = torch.tensor(cell, dtype=torch.float32)
    world = int(rng.integers(2, 7))
    kind = ["pet", "pet-adaptive", "soap"][trial % 3]
    if kind == "soap":
        fn = lambda w, r: spart.energy_and_gradient(soap, pos, z, c, pbc, w, r)
    else:
        fn = lambda w, r: ppart.energy_and_gradient(pets[kind == "pet-adaptive"], pos, z, c, pbc, w, r)
    e_ref, g_ref, _, _ = fn(1, 0)
    e_ref, g_ref = float(e_ref), g_ref.clone()
    e, grad, owned, sub = 0.0, torch.zeros_like(g_ref), 0, 0
    for r in range(world):
        er, gr, n_sub, n_owned = fn(world, r)
        e += float(er); grad += gr; owned += n_owned; sub = max(sub, n_sub)
    ee = abs(e - e_ref) / abs(e_ref)
    eg = float((grad - g_ref).abs().max() / g_ref.abs().max())
    worst = max(worst, ee, eg)
    flag = "" if owned == n and ee < 1e-5 and eg < 1e-5 else "   <-- FAIL"
    print(f"trial {trial} {kind} n={n} L={L.round(1)} pbc={pbc} world={world} busiest {sub / n:.2f}: E {ee:.1e} grad {eg:.1e}{flag}", flush=True)
print("worst", worst)

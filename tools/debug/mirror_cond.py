import sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import pet as opet
from metatrain_amd.pet import PETBackend
TYPES=[1,6,7,8]
dev=torch.device("cuda:0")
g=dict(np.load("tests/golden/batch_two_systems.npz")); t=lambda k: torch.tensor(g[k])
hypers = dict(opet.DEFAULT_HYPERS, system_conditioning=True, d_pet=8, d_head=8, d_node=8, d_feedforward=8, num_heads=1, num_attention_layers=1, num_gnn_layers=1, featurizer_type="feedforward")
params = opet.synthetic_params(hypers, TYPES, {"energy": 1}, 0, torch.float32)
charge, spin = torch.tensor([2, -1]), torch.tensor([3, 1])
be = PETBackend(hypers, TYPES); be.add_output("energy", {"energy": [1]}); be.load_state_dict(params, strict=True); be = be.to(dev).train()
keys = [k for k in params if k != "species_to_species_index"]
named = dict(be.named_parameters())
cells, sysidx = t("in_cells").float().to(dev), t("in_system_indices").to(dev)
def oracle(dtype):
    p = {k: (params[k] if k == "species_to_species_index" else named[k].detach().cpu().to(dtype).requires_grad_(True)) for k in params}
    a = opet.pet_atomic_energies(p, hypers, t("in_positions").to(dtype), t("in_cells").to(dtype), t("in_centers"), t("in_neighbors"), t("in_cell_shifts"), t("in_species"), t("in_system_indices").long(), "energy", charge=charge, spin_multiplicity=spin)
    l = a.sum()
    return float(l), dict(zip(keys, torch.autograd.grad(l, [p[k] for k in keys], allow_unused=True)))
for step in range(4):
    pos = t("in_positions").float().to(dev)
    batch = be.preprocess(pos, t("in_centers").to(dev), t("in_neighbors").to(dev), t("in_species").to(dev), cells, t("in_cell_shifts").to(dev), sysidx, 1.0)
    batch["charge"], batch["spin_multiplicity"], batch["system_indices"] = charge.to(dev), spin.to(dev), sysidx
    nodes, edges = be.calculate_features(batch)
    pred, _, _ = be.predict(nodes, edges, batch, cells, sysidx, ["energy"])
    loss = pred["energy"][0].sum(); loss.backward()
    l64, r64 = oracle(torch.float64); l32, r32 = oracle(torch.float32)
    errs=[]; e32=[]
    for k in keys:
        if r64[k] is None: continue
        sc=float(r64[k].abs().max()) or 1.0
        errs.append((float((named[k].grad.cpu().double()-r64[k]).abs().max())/sc, k, sc))
        e32.append((float((r32[k].double()-r64[k]).abs().max())/sc, k))
    errs.sort(reverse=True); e32.sort(reverse=True)
    print(step, "loss", float(loss), l64, "hip worst", errs[:3], "torch-fp32 worst", e32[:2])
    with torch.no_grad():
        for k in keys:
            if named[k].grad is not None:
                named[k] -= 0.01 * named[k].grad; named[k].grad.zero_()

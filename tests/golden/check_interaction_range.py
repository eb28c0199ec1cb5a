"""Run the imported REFERENCE (fp64, CPU) on a four-atom chain A - B - C - D with 4.0 A spacing (cutoff 4.5 A, two GNN
layers): is the energy of atom A a function of the position of atom D, 12 A = 2.67 cutoffs away -- beyond the
`num_gnn_layers x cutoff` = 9 A the reference declares as its interaction range (pet/model.py:1004)?
Run from the repo root in the build container:  python tests/golden/check_interaction_range.py
Writes tests/golden/reference_interaction_range.json (data only)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import nl as onl  # noqa: E402
from oracle import pet as opet  # noqa: E402

PETBackend = mg.import_reference_backend()
hyp = dict(opet.DEFAULT_HYPERS)
be, params = mg._reference_backend(PETBackend, hyp, torch.float64)
pos = torch.tensor([[0.0, 0, 0], [4.0, 0.3, 0], [8.0, -0.2, 0.4], [12.0, 0.1, -0.3]], dtype=torch.float64)
z = torch.tensor([6, 1, 8, 7])
cell = torch.zeros(1, 3, 3, dtype=torch.float64)
i, j, s, _ = onl.neighbor_list(pos.numpy(), cell[0].numpy(), [False] * 3, hyp["cutoff"])
i, j, s = torch.tensor(i), torch.tensor(j), torch.tensor(s)
sysidx = torch.zeros(4, dtype=torch.long)
assert sorted(zip(i.tolist(), j.tolist())) == [(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2)]  # a chain, nothing else
p = pos.clone().requires_grad_(True)
batch = be.preprocess(p, i, j, z, cell, s, sysidx, 1.0)
nf, ef = be.calculate_features(batch)
pred, _, _ = be.predict(nf, ef, batch, cell, sysidx, ["energy"])
atomic = pred["energy"][0][:, 0]
(g,) = torch.autograd.grad(atomic[0], p)  # d E_A / d R
ours = opet.pet_atomic_energies(params, hyp, pos.clone().requires_grad_(True), cell, i, j, s, z, sysidx)
out = {"spacing_A": 4.0, "cutoff_A": hyp["cutoff"], "num_gnn_layers": hyp["num_gnn_layers"],
       "declared_interaction_range_A": hyp["num_gnn_layers"] * hyp["cutoff"],
       "distance_A_to_D": float((pos[3] - pos[0]).norm()),
       "dE_A_dR": {k: [float(x) for x in g[n]] for n, k in enumerate("ABCD")},
       "max_abs_dE_A_dR_D": float(g[3].abs().max()), "max_abs_dE_A_dR_B": float(g[1].abs().max()),
       "oracle_agrees": float((ours.detach().ravel() - atomic.detach()).abs().max())}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(HERE, "reference_interaction_range.json"), "w"), indent=1)

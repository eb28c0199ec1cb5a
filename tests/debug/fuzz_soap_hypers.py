"""One-off sweep over SOAP-BPNN hyper-parameters (max_angular, max_radial, cutoff radius / width, legacy, hidden layers,
layernorm): energy + dE/dR against the fp64 oracle on a small box."""
import copy
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
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    hy = copy.deepcopy(osoap.DEFAULT_HYPERS)
    hy["soap"]["max_angular"] = int(rng.integers(1, 9))
    hy["soap"]["max_radial"] = int(rng.integers(2, 10))
    hy["soap"]["cutoff"] = {"radius": float(rng.choice([3.5, 5.0, 6.0])), "width": float(rng.choice([0.25, 0.5, 1.0]))}
    hy["legacy"] = bool(rng.random() < 0.5)
    hy["bpnn"]["num_hidden_layers"] = int(rng.integers(1, 7))
    hy["bpnn"]["layernorm"] = bool(rng.random() < 0.7)
    if rng.random() < 0.3:  # round 6: the mlp head (one more bias-free Linear + SiLU per centre species)
        hy["heads"] = {"energy": "mlp"}
    tag = f"L={hy['soap']['max_angular']} N={hy['soap']['max_radial']} rc={hy['soap']['cutoff']} legacy={hy['legacy']} " \
          f"hidden={hy['bpnn']['num_hidden_layers']} ln={hy['bpnn']['layernorm']} heads={hy.get('heads')}"
    try:
        params = osoap.synthetic_params(hy, 4, osoap.basis(hy)[0], 0, torch.float32)
        model = SoapBpnnHip(hy, types)
        model.load({k: v.to(dev) for k, v in params.items()})
    except Exception as e:
        print(f"trial {trial} {tag}: refused / failed at load: {str(e)[:120]}")
        continue
    n = int(rng.integers(20, 120))
    L = (n / 0.04) ** (1 / 3)
    cell = np.eye(3) * L
    pos = torch.tensor(rng.random((n, 3)) @ cell, dtype=torch.float32)
    cells = torch.tensor(cell, dtype=torch.float32)[None]
    z = torch.tensor(rng.choice(types, n))
    i, j, s, _ = onl.neighbor_list(pos.double().numpy(), cell, [True] * 3, hy["soap"]["cutoff"]["radius"])
    ci, cj, cs = torch.tensor(i, dtype=torch.int64), torch.tensor(j, dtype=torch.int64), torch.tensor(s, dtype=torch.int64).reshape(-1, 3)
    sysidx = torch.zeros(n, dtype=torch.int64)
    p64 = {k: v.double() for k, v in params.items()}
    e_ref, g_ref, a_ref = osoap.energy_and_gradient(p64, hy, types, pos.double(), cells.double(), ci, cj, cs, z, sysidx)
    g = model.graph(pos.to(dev), cells.to(dev), ci.to(dev), cj.to(dev), cs.to(dev), z.to(dev), sysidx.int().to(dev))
    atomic = model.forward(g)
    grad = model.backward(g, torch.ones_like(atomic))
    ea = float((atomic.cpu().double() - a_ref).abs().max() / a_ref.abs().max())
    eg = float((grad.cpu().double() - g_ref).abs().max() / g_ref.abs().max())
    flag = "" if ea < 1e-5 and eg < 1e-5 else "   <-- ABOVE 1e-5"
    print(f"trial {trial} {tag} n={n}: E {ea:.2e} grad {eg:.2e}{flag}", flush=True)

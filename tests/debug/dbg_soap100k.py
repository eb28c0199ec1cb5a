import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metatrain_amd import runtime as rt
from metatrain_amd.soap_bpnn import SoapBpnnHip
from oracle import pet as opet, soap as osoap
dev = torch.device("cuda:0")
hypers = dict(osoap.DEFAULT_HYPERS); n_per_l = osoap.basis(hypers)[0]
params = osoap.synthetic_params(hypers, 4, n_per_l, 0, torch.float32)
m = SoapBpnnHip(hypers, [1,6,7,8]); m.load({k: v.to(dev) for k, v in params.items()})
for n in (1000, 10000, 100000):
    pos, z, cell = opet.random_box(n, seed=3)
    pairs, _ = rt.neighbor_list(pos.to(dev), cell, [True]*3, 5.0)
    g = m.graph(pos.to(dev), cell[None].to(dev), pairs[:,0].contiguous(), pairs[:,1].contiguous(), pairs[:,2:5].contiguous(), z.to(dev), torch.zeros(n, dtype=torch.int32, device=dev))
    a = m.forward(g); g0 = m.backward(g, torch.ones_like(a)); sc = float(g0.abs().max())
    w = torch.rand(n, generator=torch.Generator().manual_seed(2)).to(dev)
    g1 = m.backward(g, w); g2 = m.backward(g, 1.0 - w)
    err = ((g1 + g2) - g0).abs().max(1).values / sc
    g1b = m.backward(g, w)
    print(f"n={n}: linearity max {float(err.max()):.2e} atoms>5e-6: {int((err>5e-6).sum())} median {float(err.median()):.2e}; deterministic {torch.equal(g1, g1b)}; half-seed check {float(((m.backward(g, 0.5*torch.ones_like(a))*2 - g0).abs().max())/sc):.2e}", flush=True)
    for key in ("soap_pair", "soap_sorted", "soap_mfma"):
        rt.config_set(key, 0)
        a2 = m.forward(g); h0 = m.backward(g, torch.ones_like(a2)); h1 = m.backward(g, w); h2 = m.backward(g, 1.0 - w)
        e2 = ((h1 + h2) - h0).abs().max(1).values / sc
        print(f"   {key}=0: linearity max {float(e2.max()):.2e}; vs default path g0 diff {float((h0-g0).abs().max())/sc:.2e}", flush=True)
        rt.config_set(key, 1)

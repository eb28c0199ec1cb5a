"""One-off randomized sweep of the training gradients (energy seeds nu + force-loss direction u) against torch's double
backward through the fp64 oracle: random small batches, mixed periodicity, 1-3 systems."""
import importlib.util
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")  # test_gpu_train imports its sibling _memo
spec = importlib.util.spec_from_file_location("tgt", "tests/test_gpu_train.py")
T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
from metatrain_amd import runtime as rt
from oracle import nl as onl
from oracle import pet as opet

dev = torch.device("cuda:0")
hypers = dict(opet.DEFAULT_HYPERS)
COND = len(sys.argv) > 3 and sys.argv[3] == "cond"  # system conditioning: random charges / multiplicities (with repeats)
if COND:
    hypers["system_conditioning"] = True
LN = len(sys.argv) > 3 and sys.argv[3] == "layernorm"  # normalization = LayerNorm with random norm weights / biases
if LN:
    hypers["normalization"] = "LayerNorm"
import os
for kv in filter(None, os.environ.get("PET_FUZZ_SET", "").split(",")):  # e.g. PET_FUZZ_SET=so_f16x3=0,wgrad_bf16=0
    rt.config_set(kv.split("=")[0], int(kv.split("=")[1]))
types = [1, 6, 7, 8]
params = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float32)
if LN:
    gen = torch.Generator().manual_seed(1)
    for k in params:
        if ".norm_" in k:
            params[k] = params[k] + 0.3 * torch.randn(params[k].shape, generator=gen)
model = rt.HipModel(hypers, types)
model.load({k: v.to(dev) for k, v in params.items()}, "energy")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l, off = [], [], [], [], [], [], [], 0
    desc = []
    for k in range(int(rng.integers(1, 4))):
        n = int(rng.integers(2, 70))
        rho = float(10 ** rng.uniform(-2.3, -1.1))
        L = max((n / rho) ** (1 / 3), 3.0)
        cell = np.eye(3) * L + (rng.uniform(-0.2, 0.2, (3, 3)) * L if rng.random() < 0.5 else 0.0)
        pbc = [bool(b) for b in rng.random(3) < 0.7]
        pos = rng.random((n, 3)) @ cell
        i, j, s, _ = onl.neighbor_list(pos, cell, pbc, hypers["cutoff"])
        if len(i) == 0 or np.bincount(i, minlength=n).max() > 46:
            continue
        desc.append((n, round(rho, 4), pbc))
        pos_l.append(torch.tensor(pos, dtype=torch.float32)); z_l.append(torch.tensor(rng.choice(types, n)))
        cell_l.append(torch.tensor(cell, dtype=torch.float32))
        i_l.append(torch.tensor(i, dtype=torch.int64) + off); j_l.append(torch.tensor(j, dtype=torch.int64) + off)
        s_l.append(torch.tensor(s, dtype=torch.int64).reshape(-1, 3)); sys_l.append(torch.full((n,), len(pos_l) - 1))
        off += n
    if not pos_l:
        continue
    inp = {"positions": torch.cat(pos_l), "cells": torch.stack(cell_l), "centers": torch.cat(i_l),
           "neighbors": torch.cat(j_l), "cell_shifts": torch.cat(s_l), "species": torch.cat(z_l),
           "system_indices": torch.cat(sys_l)}
    if COND:
        ns = len(pos_l)
        inp["charge"] = torch.tensor(rng.integers(-1, 2, ns))
        inp["spin_multiplicity"] = torch.tensor(rng.integers(1, 3, ns))
    n = off
    nu = torch.tensor(rng.uniform(-0.5, 0.5, n), dtype=torch.float32)
    u = torch.tensor(rng.normal(size=(n, 3)), dtype=torch.float32)
    ref, tan_ref, g_ref = T._oracle_second_order(params, hypers, inp, nu, u)
    graph = rt.HipGraph(model, inp["positions"].to(dev), inp["cells"].to(dev), inp["centers"].to(dev),
                        inp["neighbors"].to(dev), inp["cell_shifts"].to(dev), inp["species"].to(dev),
                        inp["system_indices"].int().to(dev))
    if COND:
        graph.set_conditioning(inp["charge"].to(dev), inp["spin_multiplicity"].to(dev), inp["system_indices"].to(dev))
    fw = rt.HipForward(model, graph, train=True)
    model.zero_grad()
    fw.forward()
    ones = torch.ones(n, device=dev)
    tan = fw.backward_train2(ones, nu.to(dev), u.to(dev), want_tangent=True)
    et = float(np.abs(tan.cpu().numpy() - tan_ref.numpy()).max() / np.abs(tan_ref.numpy()).max())
    got = model.grads()
    worst, wk = 0.0, ""
    for k, r in ref.items():
        r = r.numpy()
        if r.size == 1:
            continue
        g = got[k].cpu().numpy().astype(np.float64)
        scale = np.abs(r).max()
        rel = np.abs(g - r).max() / scale if scale > 1e-12 else np.abs(g - r).max()
        if rel > worst:
            worst, wk = rel, k
    flag = "" if et < 1e-5 and worst < 1e-5 else "   <-- ABOVE 1e-5"
    print(f"trial {trial} systems {desc} edges {len(inp['centers'])}: tangent {et:.2e} worst param grad {worst:.2e} ({wk}){flag}", flush=True)

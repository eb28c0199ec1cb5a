"""One-off randomized parity sweep (uses the oracle: lives under tests/): random sizes, densities, triclinic cells,
mixed periodicity, several systems per batch; per-atom energies and dE/dR against the fp64 oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt
from oracle import nl as onl
from oracle import pet as opet

dev = torch.device("cuda:0")
import os
if os.environ.get("PET_FUZZ_FUSED"):  # the per-atom fused attention block on every graph (default: graphs of >= 3 840 tiles)
    rt.config_set("attn_fused", 7)
for _kv in filter(None, os.environ.get("PET_FUZZ_SET", "").split(",")):  # e.g. PET_FUZZ_SET=node_split=0,dxf_fused=0
    rt.config_set(_kv.split("=")[0], int(_kv.split("=")[1]))
hypers = dict(opet.DEFAULT_HYPERS)
mode = sys.argv[3] if len(sys.argv) > 3 else "default"
if mode == "adaptive":
    hypers.update(num_neighbors_adaptive=10, adaptive_cutoff_method="solver", cutoff_width_adaptive=1.0)
elif mode == "grid":
    hypers.update(num_neighbors_adaptive=10, adaptive_cutoff_method="grid", cutoff_width_adaptive=1.0)
elif mode == "legacy":
    hypers.update(normalization="LayerNorm", transformer_type="PostLN", activation="SiLU")  # (residual: layered API, own tests)
elif mode == "layernorm":  # PreLN LayerNorm: the TRR layer kernels' LayerNorm instantiation, random norm weights / biases
    hypers.update(normalization="LayerNorm", activation=("SiLU" if int(sys.argv[1]) % 2 else "SwiGLU"))
elif mode == "cosine":
    hypers.update(cutoff_function="Cosine")
elif mode == "hypers":  # the size-independent hyper-parameters (the kernels are one instantiation of the sizes)
    _r = np.random.default_rng(int(sys.argv[1]) + 1000)
    hypers.update(num_gnn_layers=int(_r.integers(1, 4)), num_attention_layers=int(_r.integers(1, 4)),
                  cutoff=float(_r.choice([3.5, 4.5, 5.5])), cutoff_width=float(_r.choice([0.2, 0.5, 1.0])),
                  attention_temperature=float(_r.choice([0.5, 1.0, 2.0])),
                  cutoff_function=str(_r.choice(["Bump", "Cosine"])), activation=str(_r.choice(["SwiGLU", "SiLU"])))
    print({k: hypers[k] for k in ("num_gnn_layers", "num_attention_layers", "cutoff", "cutoff_width",
                                  "attention_temperature", "cutoff_function", "activation")})
print("mode", mode)
types = [1, 6, 7, 8]
if len(sys.argv) > 3 and sys.argv[3] == "species":  # many atomic types, random atomic numbers up to 118
    _r = np.random.default_rng(int(sys.argv[1]) + 2000)
    types = sorted(int(t) for t in _r.choice(np.arange(1, 119), int(_r.integers(2, 41)), replace=False))
    print("atomic_types", types)
p32 = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float32)
p64 = opet.synthetic_params(hypers, types, {"energy": 1}, 0, torch.float64)
if mode == "layernorm":
    _g = torch.Generator().manual_seed(int(sys.argv[1]))
    for k in p64:
        if ".norm_" in k:
            p64[k] = p64[k] + 0.3 * torch.randn(p64[k].shape, generator=_g, dtype=torch.float64)
            p32[k] = p64[k].float()
model = rt.HipModel(hypers, types)
model.load({k: v.to(dev) for k, v in p32.items()}, "energy")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = (0.0, 0.0, 0.0)
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    n_sys = int(rng.integers(1, 4))
    pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l, off = [], [], [], [], [], [], [], 0
    desc = []
    for k in range(n_sys):
        n = int(rng.integers(2, 220))
        rho = float(10 ** rng.uniform(-2.6, -0.95))
        L = max((n / rho) ** (1 / 3), 3.0)
        cell = np.eye(3) * L + (rng.uniform(-0.25, 0.25, (3, 3)) * L if rng.random() < 0.5 else 0.0)
        pbc = [bool(b) for b in rng.random(3) < 0.7]
        pos = rng.random((n, 3)) @ cell
        z = rng.choice(types, n)
        i, j, s, _ = onl.neighbor_list(pos, cell, pbc, hypers["cutoff"])
        if len(i) and np.bincount(i, minlength=n).max() > 120:
            continue
        desc.append((n, round(rho, 4), pbc))
        pos_l.append(torch.tensor(pos)); z_l.append(torch.tensor(z)); cell_l.append(torch.tensor(cell))
        i_l.append(torch.tensor(i, dtype=torch.int64) + off); j_l.append(torch.tensor(j, dtype=torch.int64) + off)
        s_l.append(torch.tensor(s, dtype=torch.int64).reshape(-1, 3)); sys_l.append(torch.full((n,), len(pos_l) - 1))
        off += n
    if not pos_l:
        continue
    pos, z, cells = torch.cat(pos_l), torch.cat(z_l), torch.stack(cell_l)
    i, j, s, sysidx = torch.cat(i_l), torch.cat(j_l), torch.cat(s_l), torch.cat(sys_l)
    graph = rt.HipGraph(model, pos.float().to(dev), cells.float().to(dev), i.to(dev), j.to(dev), s.to(dev), z.to(dev),
                        sysidx.int().to(dev))
    fw = rt.HipForward(model, graph)
    atomic = fw.forward().cpu().double().ravel()
    w = torch.tensor(rng.uniform(0.2, 2.0, off))
    grad, gcell = fw.backward(w.float().to(dev), want_cell_grad=True)
    grad, gcell = grad.cpu().double(), gcell.cpu().double()
    q = pos.float().double().requires_grad_(True)
    c64 = cells.float().double().requires_grad_(True)
    ref = opet.pet_atomic_energies(p64, hypers, q, c64, i, j, s, z, sysidx.long()).ravel()
    gp, gc = torch.autograd.grad((ref * w).sum(), [q, c64], allow_unused=True)
    gc = torch.zeros_like(c64) if gc is None else gc
    ec = float((gcell - gc).abs().max() / max(float(gc.abs().max()), 1e-30)) if float(gc.abs().max()) > 0 else 0.0
    ea = float((atomic - ref.detach()).abs().max() / ref.detach().abs().max())
    eg = float((grad - gp).abs().max() / max(float(gp.abs().max()), 1e-30)) if len(i) else 0.0
    worst = (max(worst[0], ea), max(worst[1], eg), max(worst[2], ec))
    flag = "" if ea < 1e-5 and eg < 1e-5 and ec < 1e-5 else f"   <-- ABOVE 1e-5 (cell {ec:.2e})"
    if ec > 1e-3 and os.environ.get("PET_FUZZ_DUMP"):
        print("   dE/dcell ours", gcell.numpy().round(8).tolist(), "reference", gc.numpy().round(8).tolist(), "shifts nonzero:", int((s != 0).any(1).sum()))
    if flag:  # yardstick: the same model evaluated by torch in fp32 on the CPU
        q32 = pos.float().requires_grad_(True)
        c32 = cells.float().requires_grad_(True)
        r32 = opet.pet_atomic_energies(p32, hypers, q32, c32, i, j, s, z, sysidx.long()).ravel()
        g32, gc32 = torch.autograd.grad((r32 * w.float()).sum(), [q32, c32], allow_unused=True)
        gc32 = torch.zeros_like(c32) if gc32 is None else gc32
        yc = float((gc32.double() - gc).abs().max() / max(float(gc.abs().max()), 1e-30))
        flag += f" (torch fp32: E {float((r32.detach().double() - ref.detach()).abs().max() / ref.detach().abs().max()):.2e}" \
                f" grad {float((g32.double() - gp).abs().max() / gp.abs().max()):.2e} cell {yc:.2e}; max|grad| {float(gp.abs().max()):.3e}" \
                f" max|dE/dcell| {float(gc.abs().max()):.3e})"
    print(f"trial {trial:3d} systems {desc} edges {len(i)}: E {ea:.2e} grad {eg:.2e}{flag}", flush=True)
print("worst", worst)

"""Round-3 randomized sweep over MODEL CONFIGURATIONS (uses the oracle: lives under tests/): random sizes (d_pet 1 .. 160,
d_node equal / double / unrelated, heads, feed-forward and head widths), depths, normalisation, activation, transformer and
featuriser types, cutoff functions, conditioning, adaptive cutoffs (inference), on small random batches: per-atom
energies, dE/dR, and the parameter gradients of the energy term and of the force-loss term against the fp64 oracle, with
torch's own fp32 arithmetic as the yardstick for whatever lands above the bar.   python tests/debug/fuzz_generic.py SEED N"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt
from oracle import nl as onl
from oracle import pet as opet

dev = torch.device("cuda:0")
import os
for _kv in filter(None, os.environ.get("PET_FUZZ_SET", "").split(",")):  # e.g. PET_FUZZ_SET=gen_f16x3=0
    rt.config_set(_kv.split("=")[0], int(_kv.split("=")[1]))
seed0, ntrial = int(sys.argv[1]), int(sys.argv[2])
types = [1, 6, 7, 8]
bad = 0
only = os.environ.get("PET_FUZZ_ONLY")            # run this trial index only
dhead_override = os.environ.get("PET_FUZZ_DHEAD")  # ... with this d_head instead of the drawn one (debugging)
for trial in range(ntrial):
    if only is not None and trial != int(only):
        continue
    rng = np.random.default_rng(seed0 * 1000 + trial)
    heads = int(rng.choice([1, 1, 2, 4, 8]))
    hd = int(rng.choice([1, 2, 3, 4, 8, 16, 20]))
    d_pet = heads * hd
    if rng.random() < 0.15:
        d_pet, heads = 128, 8    # the compiled d_pet with other widths around it
    d_node = int(rng.choice([d_pet, 2 * d_pet, max(1, d_pet + int(rng.integers(-3, 9)))]))
    hy = dict(opet.DEFAULT_HYPERS, d_pet=d_pet, num_heads=heads, d_node=d_node,
              d_feedforward=int(rng.choice([1, 3, 8, 16, 40, 2 * d_pet])), d_head=int(rng.choice([1, 5, 8, 32, d_pet])),
              num_gnn_layers=int(rng.integers(1, 4)), num_attention_layers=int(rng.integers(1, 4)),
              normalization=str(rng.choice(["RMSNorm", "LayerNorm"])), activation=str(rng.choice(["SwiGLU", "SiLU"])),
              transformer_type=str(rng.choice(["PreLN", "PostLN"])), featurizer_type=str(rng.choice(["feedforward", "residual"])),
              cutoff_function=str(rng.choice(["Bump", "Cosine"])), cutoff=float(rng.choice([3.5, 4.5, 5.5])),
              cutoff_width=float(rng.choice([0.2, 0.5, 1.0])), attention_temperature=float(rng.choice([0.5, 1.0, 2.0])),
              system_conditioning=bool(rng.random() < 0.35))
    if dhead_override:
        hy["d_head"] = int(dhead_override)
    adaptive = rng.random() < 0.3
    if adaptive:
        hy.update(num_neighbors_adaptive=float(rng.choice([4.0, 8.0])), adaptive_cutoff_method=str(rng.choice(["solver", "grid"])))
    tag = {k: hy[k] for k in ("d_pet", "num_heads", "d_node", "d_feedforward", "d_head", "num_gnn_layers", "num_attention_layers",
                              "normalization", "activation", "transformer_type", "featurizer_type", "cutoff_function",
                              "system_conditioning", "num_neighbors_adaptive")}
    try:
        p32 = opet.synthetic_params(hy, types, {"energy": 1}, trial, torch.float32)
        n_sys = int(rng.integers(1, 4))
        pos_l, z_l, cell_l, i_l, j_l, s_l, sys_l, off = [], [], [], [], [], [], [], 0
        for k in range(n_sys):
            n = int(rng.integers(2, 40))
            rho = float(10 ** rng.uniform(-2.2, -1.0))
            L = max((n / rho) ** (1 / 3), 3.0)
            cell = np.eye(3) * L + (rng.uniform(-0.2, 0.2, (3, 3)) * L if rng.random() < 0.5 else 0.0)
            pbc = [bool(b) for b in rng.random(3) < 0.7]
            pos = rng.random((n, 3)) @ cell
            i, j, s, _ = onl.neighbor_list(pos, cell, pbc, hy["cutoff"])
            pos_l.append(torch.tensor(pos)); z_l.append(torch.tensor(rng.choice(types, n))); cell_l.append(torch.tensor(cell))
            i_l.append(torch.tensor(i, dtype=torch.int64) + off); j_l.append(torch.tensor(j, dtype=torch.int64) + off)
            s_l.append(torch.tensor(s, dtype=torch.int64).reshape(-1, 3)); sys_l.append(torch.full((n,), k))
            off += n
        pos, z, cells = torch.cat(pos_l).float(), torch.cat(z_l), torch.stack(cell_l).float()
        i, j, s, sysidx = torch.cat(i_l), torch.cat(j_l), torch.cat(s_l), torch.cat(sys_l)
        if len(i) == 0:
            continue
        kw = {}
        if hy["system_conditioning"]:
            kw = dict(charge=torch.tensor(rng.integers(-3, 4, n_sys)), spin_multiplicity=torch.tensor(rng.integers(1, 5, n_sys)))
        model = rt.HipModel(hy, types)
        model.load({k: v.to(dev) for k, v in p32.items()}, "energy")
        graph = rt.HipGraph(model, pos.to(dev), cells.to(dev), i.to(dev), j.to(dev), s.to(dev), z.to(dev), sysidx.int().to(dev))
        if kw:
            graph.set_conditioning(kw["charge"].to(dev), kw["spin_multiplicity"].to(dev), sysidx.to(dev))

        def oracle(dtype, what):
            p = {k: (v if k == "species_to_species_index" else v.to(dtype).clone().requires_grad_(True)) for k, v in p32.items()}
            q = pos.to(dtype).requires_grad_(True)
            a = opet.pet_atomic_energies(p, hy, q, cells.to(dtype), i, j, s, z, sysidx.long(), "energy", **kw)[:, 0]
            keys = [k for k in p if k != "species_to_species_index"]
            if what == "infer":
                (g,) = torch.autograd.grad((a * w.to(dtype)).sum(), q)
                return a.detach(), g
            if what == "energy":
                gr = torch.autograd.grad((a * w.to(dtype)).sum(), [p[k] for k in keys], allow_unused=True)
            else:
                (g,) = torch.autograd.grad(a.sum(), q, create_graph=True)
                gr = torch.autograd.grad((nu.to(dtype) * a).sum() + (u.to(dtype) * g).sum(), [p[k] for k in keys], allow_unused=True)
            return {k: (torch.zeros_like(p[k]) if x is None else x.detach()) for k, x in zip(keys, gr)}

        w = torch.tensor(rng.uniform(0.2, 2.0, off))
        nu = torch.tensor(rng.uniform(-0.5, 0.5, off))
        u = torch.tensor(rng.normal(size=(off, 3)))
        msgs = []

        def check(name, got, ref, ref32, tol):
            nonlocal_bad = 0
            sc = float(ref.abs().max())
            err = float((got.cpu().double() - ref).abs().max()) / (sc if sc > 1e-12 else 1.0)
            if not err < tol:
                e32 = float((ref32.double() - ref).abs().max()) / (sc if sc > 1e-12 else 1.0)
                msgs.append(f"{name} {err:.2e} (torch fp32 {e32:.2e}; max|ref| {sc:.3e}, numel {ref.numel()})")
                if err > 5 * max(e32, 2e-6):
                    nonlocal_bad = 1
            return nonlocal_bad

        # inference
        fw = rt.HipForward(model, graph)
        a = fw.forward()
        g = fw.backward(w.float().to(dev))
        a64, g64 = oracle(torch.float64, "infer")
        a32, g32 = oracle(torch.float32, "infer")
        flagged = check("E", a, a64, a32, 1e-5) + check("dE/dR", g, g64, g32, 1e-5)
        # training (the force-loss pass carries the cutoff tangents of the 'solver' method only)
        trainable = not adaptive or hy["adaptive_cutoff_method"] == "solver"
        if trainable:
            ft = rt.HipForward(model, graph, train=True)
            for what in ("energy", "force"):
                model.zero_grad()
                ft.forward()
                if what == "energy":
                    ft.backward_train(w.float().to(dev))
                else:
                    ft.backward(torch.ones(off, device=dev))
                    ft.backward_train2(torch.ones(off, device=dev), nu.float().to(dev), u.float().to(dev))
                got = model.grads()
                r64, r32 = oracle(torch.float64, what), oracle(torch.float32, what)
                for k, r in r64.items():
                    gk = got[k].cpu().double()
                    if gk.shape != r.shape:   # SiLU: [W; W] halves
                        gk = gk[: r.shape[0]] + gk[r.shape[0]:]
                    flagged += check(f"{what}:{k}", gk, r, r32[k], 2e-5)
        status = "BAD" if flagged else ("above bar, within the fp32 yardstick" if msgs else "ok")
        bad += 1 if flagged else 0
        print(f"trial {trial:3d} atoms {off} edges {len(i)} {status} {tag}" + ("".join("\n      " + m for m in (msgs if flagged else msgs[:6]))), flush=True)
    except Exception as exc:   # noqa: BLE001
        bad += 1
        print(f"trial {trial:3d} EXCEPTION {type(exc).__name__}: {str(exc)[:300]} {tag}", flush=True)
print("bad trials:", bad)

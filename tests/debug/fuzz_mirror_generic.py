"""Round-3 sweep of the torch mirror (``metatrain_amd.pet.PETBackend``) over random model configurations: strict
``load_state_dict`` of the oracle's (= the reference's) key schema, eager energies and dE/dR, TorchScript round trip, and
``train()``-mode ``loss.backward()`` with a force term against the fp64 oracle.   python tests/debug/fuzz_mirror_generic.py SEED N"""
import io
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from metatrain_amd.pet import PETBackend
from oracle import nl as onl
from oracle import pet as opet

dev = torch.device("cuda:0")
seed0, ntrial = int(sys.argv[1]), int(sys.argv[2])
types = [1, 6, 7, 8]
bad = 0
for trial in range(ntrial):
    rng = np.random.default_rng(seed0 * 7919 + trial)
    heads = int(rng.choice([1, 2, 4, 8]))
    hd = int(rng.choice([2, 3, 4, 8, 16]))
    d_pet = heads * hd
    hy = dict(opet.DEFAULT_HYPERS, d_pet=d_pet, num_heads=heads, d_node=int(rng.choice([d_pet, 2 * d_pet, d_pet + 5])),
              d_feedforward=int(rng.choice([4, 8, 16, 2 * d_pet])), d_head=int(rng.choice([4, 8, 32])),
              num_gnn_layers=int(rng.integers(1, 4)), num_attention_layers=int(rng.integers(1, 3)),
              normalization=str(rng.choice(["RMSNorm", "LayerNorm"])), activation=str(rng.choice(["SwiGLU", "SiLU"])),
              transformer_type=str(rng.choice(["PreLN", "PostLN"])), featurizer_type=str(rng.choice(["feedforward", "residual"])),
              cutoff_function=str(rng.choice(["Bump", "Cosine"])), system_conditioning=bool(rng.random() < 0.4))
    tag = {k: hy[k] for k in ("d_pet", "num_heads", "d_node", "d_feedforward", "d_head", "num_gnn_layers", "num_attention_layers",
                              "normalization", "activation", "transformer_type", "featurizer_type", "system_conditioning")}
    try:
        p32 = opet.synthetic_params(hy, types, {"energy": 1}, trial, torch.float32)
        n = int(rng.integers(3, 40))
        L = (n / 0.05) ** (1 / 3)
        cell = np.eye(3) * L
        pos = rng.random((n, 3)) @ cell
        i, j, s, _ = onl.neighbor_list(pos, cell, [True] * 3, hy["cutoff"])
        z = torch.tensor(rng.choice(types, n))
        ti = lambda a: torch.tensor(np.asarray(a), dtype=torch.long)  # noqa: E731
        I, J, S = ti(i), ti(j), ti(s).reshape(-1, 3)
        sysidx = torch.zeros(n, dtype=torch.long)
        cells = torch.tensor(cell, dtype=torch.float32)[None]
        kw = dict(charge=torch.tensor([int(rng.integers(-3, 4))]), spin_multiplicity=torch.tensor([int(rng.integers(1, 5))])) \
            if hy["system_conditioning"] else {}
        be = PETBackend(hy, types)
        be.add_output("energy", {"energy": [1]})
        be.load_state_dict(p32, strict=True)
        be = be.to(dev)

        def run(b, train):
            b = b.train() if train else b.eval()
            p = torch.tensor(pos, dtype=torch.float32, device=dev).requires_grad_(True)
            batch = b.preprocess(p, I.to(dev), J.to(dev), z.to(dev), cells.to(dev), S.to(dev), sysidx.to(dev), 1.0)
            if kw:
                batch["charge"], batch["spin_multiplicity"], batch["system_indices"] = kw["charge"].to(dev), kw["spin_multiplicity"].to(dev), sysidx.to(dev)
            nodes, edges = b.calculate_features(batch)
            pred, _, _ = b.predict(nodes, edges, batch, cells.to(dev), sysidx.to(dev), ["energy"])
            a = pred["energy"][0][:, 0]
            (g,) = torch.autograd.grad(a.sum(), p, create_graph=train)
            return a, g

        p64 = {k: (v if k == "species_to_species_index" else v.double().clone().requires_grad_(True)) for k, v in p32.items()}
        q = torch.tensor(pos, dtype=torch.float32).double().requires_grad_(True)
        a64 = opet.pet_atomic_energies(p64, hy, q, cells.double(), I, J, S, z, sysidx, "energy", **kw)[:, 0]
        (g64,) = torch.autograd.grad(a64.sum(), q, create_graph=True)
        tgt = torch.tensor(rng.normal(size=(n, 3)) * 0.3)
        we = torch.tensor(rng.uniform(-0.5, 0.5, n))
        l64 = (we * a64).sum() + ((g64 - tgt) ** 2).sum()
        keys = [k for k in p64 if k != "species_to_species_index"]
        r64 = dict(zip(keys, torch.autograd.grad(l64, [p64[k] for k in keys], allow_unused=True)))
        msgs = []
        a, g = run(be, False)
        ea = float((a.detach().cpu().double() - a64.detach()).abs().max() / a64.detach().abs().max())
        eg = float((g.cpu().double() - g64.detach()).abs().max() / g64.detach().abs().max())
        buf = io.BytesIO()
        torch.jit.save(torch.jit.script(be.eval()), buf)
        buf.seek(0)
        a2, g2 = run(torch.jit.load(buf, map_location=dev), False)
        es = float((a2 - a).abs().max()) + float((g2 - g).abs().max())
        a, g = run(be, True)
        loss = (we.float().to(dev) * a).sum() + ((g - tgt.float().to(dev)) ** 2).sum()
        loss.backward()
        named = dict(be.named_parameters())
        worst, wk = 0.0, ""
        for k in keys:
            if r64[k] is None:
                continue
            sc = float(r64[k].abs().max())
            e = float((named[k].grad.cpu().double() - r64[k]).abs().max()) / (sc if sc > 1e-12 else 1.0)
            if e > worst:
                worst, wk = e, k
        ok = ea < 1e-5 and eg < 1e-5 and es == 0.0 and worst < 5e-5
        bad += 0 if ok else 1
        print(f"trial {trial:3d} n {n} E {ea:.1e} dE/dR {eg:.1e} scripted-eager {es:.1e} grads {worst:.1e} ({wk}) {'ok' if ok else 'CHECK'} {tag}", flush=True)
    except Exception as exc:   # noqa: BLE001
        bad += 1
        print(f"trial {trial:3d} EXCEPTION {type(exc).__name__}: {str(exc)[:400]} {tag}", flush=True)
print("flagged trials:", bad)

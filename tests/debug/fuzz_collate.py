"""One-off randomized sweep of the batched device neighbour list (data.collate -> pet_nl_build_batch): 1-6 systems per
batch, random cells / periodicity, pair sets against the oracle per system."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from metatrain_amd import data
from oracle import nl as onl

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = compared = 0
for t in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
    cutoff = float(rng.choice([3.5, 4.5, 6.0]))
    systems, ref, off, skip = [], set(), 0, False
    for k in range(int(rng.integers(1, 7))):
        n = int(rng.integers(1, 200))
        L = rng.uniform(3.0, 22.0, 3)
        cell = np.diag(L) + (rng.uniform(-0.25, 0.25, (3, 3)) * L.min() if rng.random() < 0.5 else 0.0)
        pbc = tuple(bool(b) for b in rng.random(3) < 0.65)
        if not any(pbc):
            cell = np.zeros((3, 3))
            pos = rng.random((n, 3)) * L
        else:
            pos = (rng.random((n, 3)) * 1.3 - 0.15) @ cell
        if n > 80 and min(L) < cutoff:
            n = 80; pos = pos[:80]
        p32, c32 = torch.tensor(pos, dtype=torch.float32), torch.tensor(cell, dtype=torch.float32)
        i, j, s, d = onl.neighbor_list(p32.double().numpy(), c32.double().numpy(), list(pbc), cutoff)
        if len(i) and np.any(np.abs(np.linalg.norm(d, axis=1) - cutoff) < 2e-5 * cutoff):
            skip = True
        ref |= {(int(a) + off, int(b) + off, *map(int, sh)) for a, b, sh in zip(i, j, s)}
        systems.append((p32.to(dev), torch.ones(n, dtype=torch.int64, device=dev), c32.to(dev), pbc))
        off += n
    if skip:
        continue
    batch = data.collate(systems, cutoff, {})
    got = torch.cat([batch["centers"][:, None], batch["neighbors"][:, None], batch["cell_shifts"]], 1).cpu().tolist()
    compared += 1
    if not (len(got) == len(ref) and {tuple(r) for r in got} == ref):
        bad += 1
        print(f"trial {t}: MISMATCH systems {[(len(s[1]), s[3]) for s in systems]} cutoff {cutoff}: {len(got)} vs {len(ref)}")
print("compared", compared, "mismatches", bad)

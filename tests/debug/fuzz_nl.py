"""One-off randomized sweep of the device neighbour list against the oracle (sorted (i, j, S) sets + vectors)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from metatrain_amd import runtime as rt
from oracle import nl as onl

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
compared = pairs_total = 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for t in range(trials):
    n = int(rng.integers(1, 400))
    cutoff = float(rng.choice([3.0, 4.5, 5.0, 6.5]))
    L = rng.uniform(2.5, 25.0, 3)
    cell = np.diag(L)
    if rng.random() < 0.6:
        cell = cell + rng.uniform(-0.3, 0.3, (3, 3)) * L.min()
    pbc = [bool(b) for b in rng.random(3) < 0.65]
    if n > 150 and min(L) < cutoff:  # keep the pair count bounded
        n = 150
    pos = (rng.random((n, 3)) * rng.uniform(0.8, 1.6) - rng.uniform(0, 0.3)) @ cell  # partly outside the cell
    pos32 = torch.tensor(pos, dtype=torch.float32)
    cell32 = torch.tensor(cell, dtype=torch.float32)
    try:
        i, j, s, d = onl.neighbor_list(pos32.double().numpy(), cell32.double().numpy(), pbc, cutoff)
    except Exception as e:  # the oracle's own limits
        print(f"trial {t}: oracle failed ({e}) n={n} pbc={pbc}")
        continue
    dist = np.linalg.norm(d, axis=1) if len(i) else np.zeros(0)
    if len(i) and np.any(np.abs(dist - cutoff) < 2e-5 * cutoff):
        continue  # a pair sitting on the cutoff in fp32 vs fp64: set membership is rounding
    pairs, vec = rt.neighbor_list(pos32.to(dev), cell32, pbc, cutoff)
    got = pairs.cpu().numpy()
    compared += 1
    pairs_total += len(i)
    ok = len(got) == len(i)
    if ok and len(i):
        order = np.lexsort((got[:, 4], got[:, 3], got[:, 2], got[:, 1], got[:, 0]))
        ok = np.array_equal(got[order], np.column_stack([i, j, s])) and np.all(np.diff(got[:, 0]) >= 0)
        ok = ok and np.abs(vec.cpu().numpy()[order] - d).max() < 5e-5 * max(1.0, np.abs(pos).max())
    if not ok:
        bad += 1
        print(f"trial {t}: MISMATCH n={n} cutoff={cutoff} pbc={pbc} L={L.round(2)} pairs {len(got)} vs {len(i)}")
print("trials", trials, "compared", compared, "pairs", pairs_total, "mismatches", bad)

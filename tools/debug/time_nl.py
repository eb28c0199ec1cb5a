import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metatrain_amd import runtime as rt
from metatrain_amd.synthetic import random_box
dev = torch.device("cuda:0")
for n_box, n in ((1, 10000), (8, 10000), (64, 1000), (1, 100000)):
    boxes = [random_box(n, seed=b) for b in range(n_box)]
    pos = torch.cat([b[0] for b in boxes]).to(dev); cells = torch.stack([b[2] for b in boxes])
    first = [k * n for k in range(n_box + 1)]
    for _ in range(3):
        pairs, _ = rt.neighbor_list_batch(pos, cells, [[True] * 3] * n_box, first, 4.5, want_vectors=False)
    torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 10
    for _ in range(reps):
        pairs, _ = rt.neighbor_list_batch(pos, cells, [[True] * 3] * n_box, first, 4.5, want_vectors=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{n_box} x {n} atoms: {len(pairs)} pairs, {dt*1e3:.3f} ms per batch = {dt*1e3/n_box:.3f} ms per box", flush=True)

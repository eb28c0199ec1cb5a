import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metatrain_amd import runtime as rt
from oracle import pet as opet
dev = torch.device("cuda:0")
hypers = dict(opet.DEFAULT_HYPERS)
params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
model = rt.HipModel(hypers, [1, 6, 7, 8]); model.load({k: v.to(dev) for k, v in params.items()}, "energy")
lib = rt._lib.load()
E = 191044
gen = torch.Generator(device=dev).manual_seed(0)
dXe = torch.randn(E, 128, device=dev, generator=gen) * 1e-2
a0 = torch.randn(E, 128, device=dev, generator=gen)
wct = torch.empty(4, 128, device=dev)
for rep in range(4):
    dgeo = torch.zeros(E, 4, device=dev); da0 = torch.zeros(E, 128, device=dev); dbg = torch.zeros(E, 3, 2, 4, device=dev)
    rc = lib.pet_debug_compress_bwd(model.handle, rt._ptr(dXe), rt._ptr(a0), rt._ptr(dgeo), rt._ptr(da0), rt._ptr(dbg), ctypes.c_int64(E), rt._ptr(wct), rt._stream())
    assert rc == 0
    torch.cuda.synchronize()
    want = (da0.double() @ wct.double().T)                     # [E,4] from the kernel's own da0
    part = dbg[:, 0].double()                                   # [E,2,4] partial sums per half
    # expected partials: half h owns features 8kg+4h..+3
    cols = torch.arange(128, device=dev).reshape(16, 2, 4)     # [kg][h][j]
    wp = torch.stack([(da0.double()[:, cols[:, h].reshape(-1)] @ wct.double()[:, cols[:, h].reshape(-1)].T) for h in range(2)], 1)  # [E,2,4]
    sc = float(want.abs().max())
    bad_final = ((dgeo.double() - want).abs().max(1).values > 1e-4 * sc).nonzero().flatten()
    bad_part = ((part - wp).abs().amax((1, 2)) > 1e-4 * sc).nonzero().flatten()
    summed = dbg[:, 1].double()
    bad_sum = ((summed - part.sum(1, keepdim=True)).abs().amax((1, 2)) > 1e-4 * sc).nonzero().flatten()
    old = dbg[:, 2, 0]
    print(f"rep{rep}: bad final rows {len(bad_final)}  bad partial rows {len(bad_part)}  bad cross-lane-sum rows {len(bad_sum)}  nonzero old {int((old != 0).any(1).sum())}", flush=True)
    if len(bad_final):
        r = int(bad_final[0])
        print("  row", r, "dgeo", dgeo[r].tolist(), "\n   want", want[r].tolist(), "\n   partials", part[r].tolist(), "\n   want partials", wp[r].tolist(), "\n   summed", summed[r].tolist(), flush=True)
    # which term explains the difference?
    stats = {}
    for r in bad_part[:40].tolist():
        d = (part[r] - wp[r])                       # [2,4]
        h, q = divmod(int(d.abs().argmax()), 4)
        feats = cols[:, h].reshape(-1)              # 64 features of this half, order kg-major
        terms = da0[r, feats].double() * wct[q, feats].double()
        diff = float(d[h, q])
        k = int((terms + diff).abs().argmin())      # missing term: diff = -term
        k2 = int((terms - diff).abs().argmin())     # doubled term: diff = +term
        stats.setdefault((h, q), []).append((k, float(terms[k]), diff, k2, float(terms[k2])))
    for (h, q), v in stats.items():
        print(f"  half {h} comp {q}: {len(v)} rows; (best 'missing' feature idx, its term, diff, best 'doubled' idx, term): {v[:4]}", flush=True)

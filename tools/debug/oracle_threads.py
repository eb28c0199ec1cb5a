"""How many torch CPU threads should the GPU suite's oracle calls use on this host?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_train as T
from oracle import pet as opet
print("cores", os.cpu_count(), "default threads", torch.get_num_threads())
hypers = dict(opet.DEFAULT_HYPERS)
params = opet.synthetic_params(hypers, [1, 6, 7, 8], {"energy": 1}, 0, torch.float32)
inp = T._inputs(os.path.join(ROOT, "tests", "golden"), "batch_two_systems.npz")
n = inp["positions"].shape[0]
gen = torch.Generator().manual_seed(3)
nu = torch.rand(n, generator=gen) - 0.5
u = torch.randn(n, 3, generator=gen)
for th in (0, 64, 32, 16, 8, 4):
    if th: torch.set_num_threads(th)
    t0 = time.time()
    T._oracle_second_order.__wrapped__(params, hypers, inp, nu, u)
    print("threads", th or "default", f"{time.time() - t0:.2f} s", flush=True)

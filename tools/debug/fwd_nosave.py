"""Stage times of the forward with and without saved activations (save_for_backward = 1 | 0) at 8 x 10k atoms."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
exec(open(os.path.join(ROOT, "tools/debug/ablk_split_run.py")).read().split("rt.config_set(\"side_stream\"")[0])
from metatrain_amd.runtime import _ptr, _stream, check
rt.config_set("side_stream", 0)
fw = rt.HipForward(model, graph)
atomic = torch.empty(nb * n, device=dev)
for save in (1, 0):
    def f():
        check(fw.lib.pet_forward(model.handle, graph.handle, _ptr(fw.workspace), fw.nbytes, save, _ptr(atomic), None, None, _stream()))
    f(); torch.cuda.synchronize()
    rt.profile(True)
    for _ in range(5): f()
    torch.cuda.synchronize(); rep = rt.profile_report(); rt.profile(False)
    print(f"save={save}: " + ", ".join(f"{r['name']} {r['total_ms'] / r['calls']:.3f}" for r in rep if r['total_ms'] / r['calls'] > 0.3), flush=True)

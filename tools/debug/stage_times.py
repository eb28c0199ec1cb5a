"""Per-stage times (single stream) of forward + adjoint at 8 x 10k atoms."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
exec(open(os.path.join(ROOT, "tools/debug/ablk_split_run.py")).read().split("rt.config_set(\"side_stream\"")[0])
rt.config_set("side_stream", 0)
fw = rt.HipForward(model, graph)
ones = torch.ones(nb * n, device=dev)
for _ in range(2):
    fw.forward(); fw.backward(ones)
torch.cuda.synchronize()
rt.profile(True)
for _ in range(5):
    fw.forward(); fw.backward(ones)
torch.cuda.synchronize(); rep = rt.profile_report(); rt.profile(False)
tot = sum(r["total_ms"] for r in rep) / 5
print(os.environ.get("TAG", ""), f"sum {tot:.2f} ms |", ", ".join(f"{r['name']} {r['total_ms'] / r['calls']:.3f}" for r in sorted(rep, key=lambda r: -r['total_ms'])[:12]), flush=True)

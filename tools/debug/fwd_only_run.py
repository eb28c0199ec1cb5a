"""8 x 10k atoms, forward (save = 1) a few times (for rocprofv3 passes of the forward kernels)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
exec(open(os.path.join(ROOT, "tools/debug/ablk_split_run.py")).read().split("rt.config_set(\"side_stream\"")[0])
rt.config_set("side_stream", 0)
for kv in os.environ.get("SET", "").split(","):
    if kv: rt.config_set(kv.split("=")[0], int(kv.split("=")[1]))
fw = rt.HipForward(model, graph)
for _ in range(3): fw.forward()
torch.cuda.synchronize()

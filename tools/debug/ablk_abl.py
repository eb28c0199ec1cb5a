"""Timing ablations of k_ablk_bwd_core (pet_config_set("attn_bwd_abl", bits)); results are wrong by design."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
exec(open(os.path.join(ROOT, "tools/debug/ablk_split_run.py")).read().split("rt.config_set(\"side_stream\"")[0])
rt.config_set("side_stream", 0)
fw = rt.HipForward(model, graph)
ones = torch.ones(nb * n, device=dev)
for abl in [int(x) for x in os.environ.get("ABL", "0,1,2,4,8,15").split(",")]:
    rt.config_set("attn_bwd_abl", abl)
    fw.forward(); fw.backward(ones); torch.cuda.synchronize()
    rt.profile(True)
    for _ in range(3):
        fw.forward(); fw.backward(ones)
    torch.cuda.synchronize(); rep = rt.profile_report(); rt.profile(False)
    for r in rep:
        if r["name"] == "attn_blk_bwd":
            print(f"abl={abl:2d} attn_blk_bwd {r['total_ms'] / r['calls']:8.3f} ms per launch (core + x)", flush=True)
rt.config_set("attn_bwd_abl", 0)

import sys, torch
sys.path.insert(0, ".")
exec(open("tools/gpu_train_bench.py").read().split("def timeit")[0])
u = torch.randn(boxes * natoms, 3, device=dev) * 1e-3
fw.forward(); fw.backward_train2(seeds, seeds, u); torch.cuda.synchronize()
rt.profile(True)
fw.backward_train2(seeds, seeds, u)
torch.cuda.synchronize()
tot = 0
for r in sorted(rt.profile_report(), key=lambda r: -r["total_ms"]):
    tot += r["total_ms"]
    print("  %-18s %8.3f ms x%-3d %8.1f TF/s %8.0f GB/s" % (r["name"], r["total_ms"], r["calls"], r["flops"] / max(r["total_ms"], 1e-9) / 1e9, r["bytes"] / max(r["total_ms"], 1e-9) / 1e6))
print("sum", tot)

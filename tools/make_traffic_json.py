"""profiles/<tag>_rocprofv3_summary.txt (FETCH_SIZE / WRITE_SIZE sections) -> profiles/r0N_traffic.json (argv[3]).
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE counts 64 B units
reported in KiB on gfx950, i.e. half the bytes; WRITE_SIZE in KiB)."""
import json
import re
import sys

src, edges = sys.argv[1], int(sys.argv[2])
dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/r02_traffic.json"
step_kernel = sys.argv[4] if len(sys.argv) > 4 else "k_head<256>"   # a kernel launched once per step: its calls = steps profiled
command = sys.argv[5] if len(sys.argv) > 5 else "bench.py, default workload"
text = open(src).read()
sections = text.split("# counters")
vals = {}
for sec in sections[1:]:
    lines = sec.splitlines()
    header = lines[1].split()
    if header[-1] not in ("FETCH_SIZE", "WRITE_SIZE"):
        continue
    for l in lines[2:]:
        parts = l.split()
        if len(parts) < 3:
            continue
        try:
            v = float(parts[-1]); ncalls = int(parts[-2])
        except ValueError:
            continue
        name = " ".join(parts[:-2])
        vals.setdefault(name, {})[header[-1]] = v
        vals[name]["calls"] = ncalls
out = {"source": f"{src} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, {command}); "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md",
       "workload_edges": edges, "kernels": {}}
for name, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and name.startswith("k_"):
        out["kernels"][name] = {"calls": v["calls"], "fetch_kib_raw": v["FETCH_SIZE"], "write_kib_raw": v["WRITE_SIZE"],
                                "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024}
# whole-step traffic over ALL kernels; `step_kernel` (PET: the node head) runs once per step, so its call count is the number of steps profiled
steps = max([v["calls"] for k, v in out["kernels"].items() if k.startswith(step_kernel)] or [1])
out["steps_profiled"] = steps
out["step_hbm_bytes"] = sum(v["hbm_bytes_per_launch"] * v["calls"] for v in out["kernels"].values()) / steps
json.dump(out, open(dst, "w"), indent=1)
print(len(out["kernels"]), "kernels")

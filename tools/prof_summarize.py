"""Reduce rocprofv3 CSV output (kernel trace / counter collection) to small per-kernel tables."""
import csv, glob, os, re, sys
from collections import defaultdict

def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("pet::", "")
    return name[:48]

def main(root, out):
    lines = []
    for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot = sum(v[1] for v in agg.values())
        lines.append(f"# kernel trace: {os.path.relpath(f, root)}  (total {tot/1e3:.3f} ms of kernel time)")
        lines.append(f"{'kernel':50s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
        for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"{k:50s} {n:7d} {us:12.1f} {us/n:10.2f} {100*us/tot:6.2f}")
        lines.append("")
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
        names = sorted({c for v in agg.values() for c in v})
        lines.append(f"# counters (per-dispatch average): {os.path.relpath(f, root)}")
        lines.append(f"{'kernel':50s} {'calls':>6s} " + " ".join(f"{c[-26:]:>26s}" for c in names))
        for k in sorted(agg, key=lambda k: -max(agg[k].values())):
            n = len(calls[k])
            lines.append(f"{k:50s} {n:6d} " + " ".join(f"{agg[k].get(c, 0.0)/n:26.1f}" for c in names))
        lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

"""gpurun_out/trace_rows.csv (tools/debug/trace_gaps.sh) -> per step: wall time, union of busy time, idle gaps, time with
two kernels in flight, and the largest gaps with the kernels either side."""
import csv, sys
rows = []
for line in list(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trace_rows.csv"))[1:]:
    name, s, e, q = line.rstrip("\n").rsplit(",", 3)   # kernel names hold commas
    rows.append((name, int(s), int(e), q))
rows.sort(key=lambda r: r[1])
# steps: delimited by k_edge_geometry (once per graph build)
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_edge_geometry")]
for a, b in zip(starts[-4:-1], starts[-3:]):
    seg = rows[a:b]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    t1 = rows[b][1]
    ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
    busy = over = 0; depth = 0; last = t0; gaps = []
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
        if depth == 0 and t > last: gaps.append((t - last, t))
        depth += d; last = t
    ksum = sum(r[2] - r[1] for r in seg)
    print(f"step: wall {1e-6*(t1-t0):.2f} ms, busy {1e-6*busy:.2f}, kernel-time sum {1e-6*ksum:.2f}, >=2 in flight {1e-6*over:.2f}, idle {1e-6*(t1-t0-busy):.2f} in {len(gaps)} gaps")
    gaps.sort(reverse=True)
    for g, t in gaps[:8]:
        before = max((r for r in seg if r[2] <= t - g + 1), key=lambda r: r[2], default=None)
        after = min((r for r in seg if r[1] >= t), key=lambda r: r[1], default=None)
        print(f"   gap {g/1e3:7.1f} us  after {before[0] if before else '-':28s} before {after[0] if after else '-'}")
seg = rows[starts[-2]:starts[-1]]
print("queues:", {q: sum(1 for r in seg if r[3] == q) for q in set(r[3] for r in seg)})

#!/bin/bash
# kernel trace of the single-box step: per-kernel table + kernel-time sum against the step time
export TMPDIR=/tmp
RAW=/tmp/prof_box1; rm -rf $RAW; mkdir -p $RAW gpurun_out
B=${BOXES:-1}
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python bench.py --boxes $B --steps 20 --warmup 5 --no-cpu-baseline > $RAW/trace.log 2>&1
tail -1 $RAW/trace.log > gpurun_out/box${B}_traced.json
python tools/prof_summarize.py $RAW gpurun_out/box${B}_trace.txt > /dev/null
python - <<PY
import csv, glob
f = glob.glob("$RAW/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last 10 steps: find the span of the final 10 'k_geom' launches (one per graph build)
idx = [i for i, r in enumerate(rows) if "k_pos_grad" in r["Kernel_Name"]]
a, b = idx[-11] + 1, idx[-1] + 1
seg = rows[a:b]
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
# union of intervals (two streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print("10 steps: span %.3f ms/step, kernel sum %.3f ms/step, union(busy) %.3f ms/step, launches/step %d" % (span/1e7, busy/1e7, u/1e7, len(seg)//10))
PY

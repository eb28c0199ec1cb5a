#!/bin/bash
# raw kernel trace of the 1000-atom MD probe (tools/debug/trace_gaps.py reduces it)
export TMPDIR=/tmp
RAW=/tmp/prof_b1k
rm -rf $RAW; mkdir -p $RAW gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $RAW/trace -o trace -- python tools/gpu_md_probe.py 1000 > $RAW/trace.log 2>&1
tail -2 $RAW/trace.log
f=$(find $RAW -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows = rows[-1500:]
out = open("gpurun_out/trace_rows_b1k.csv", "w")
out.write("name,start,end,queue\n")
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("pet::", "")[:40]
    out.write(f"{n},{r['Start_Timestamp']},{r['End_Timestamp']},{r.get('Queue_Id','')}\n")
PY

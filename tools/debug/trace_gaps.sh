#!/bin/bash
# kernel-trace of the bench with raw per-dispatch timestamps kept (tools/debug/trace_gaps.py reduces them to busy / idle time per step)
export TMPDIR=/tmp
RAW=/tmp/prof_gaps
rm -rf $RAW; mkdir -p $RAW gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $RAW/trace -o trace -- python ${BENCH_PY:-bench.py} --steps 4 --warmup 2 --no-cpu-baseline ${BENCH_EXTRA---no-extras} $BENCH_ARGS > $RAW/trace.log 2>&1
tail -2 $RAW/trace.log
f=$(find $RAW -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
out = open("gpurun_out/trace_rows.csv", "w")
out.write("name,start,end,queue\n")
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("pet::", "")[:40]
    out.write(f"{n},{r['Start_Timestamp']},{r['End_Timestamp']},{r.get('Queue_Id','')}\n")
print(len(rows), "dispatches")
PY

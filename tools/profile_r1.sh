#!/bin/bash
# rocprofv3 evidence for the bench configuration; only reduced tables leave the box.
export TMPDIR=/tmp
RAW=/tmp/prof_raw
OUT=$PWD/gpurun_out
rm -rf $RAW; mkdir -p $RAW $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $RAW/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES -d $RAW/pmc1 -o pmc1 -- $CMD > $RAW/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $RAW/pmc2 -o pmc2 -- $CMD > $RAW/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $RAW/pmc3 -o pmc3 -- $CMD > $RAW/pmc3.log 2>&1
python tools/prof_summarize.py $RAW $OUT/prof_summary.txt > /dev/null
cp $(find $RAW -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python - <<'PY'
rows = [l for l in open("gpurun_out/prof_summary.txt").read().split("# counters")[1].splitlines()[1:] if l.strip()]
print("kernel".ljust(30), "calls  us/call  mfma_util  wait_any%  wait_inst%  active%   waves  kcyc/wave")
for l in rows[1:45]:
    parts = l.split()
    try:
        vals = list(map(float, parts[-8:]))
    except ValueError:
        continue
    name = " ".join(parts[:-9]); calls = parts[-9]
    gui, active, busy, mfma, wait_any, wait_inst, waves, wcyc = vals
    if wcyc == 0: continue
    util = (mfma / 1024) / (gui / 8) if gui else 0
    print(name[:30].ljust(30), calls.rjust(5), f"{gui/8/2.3e3:8.0f}", f"{util:9.2f}", f"{100*wait_any/wcyc:9.1f}", f"{100*wait_inst/wcyc:10.1f}", f"{100*active/wcyc:8.1f}", f"{waves:7.0f}", f"{4*wcyc/waves/1e3:9.1f}")
PY

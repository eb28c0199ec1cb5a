#!/bin/bash
# PMC pass 1 only (SQ counters) on the bench; reduced table printed and saved.
export TMPDIR=/tmp
RAW=/tmp/prof_raw
OUT=$PWD/gpurun_out
rm -rf $RAW; mkdir -p $RAW $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES -d $RAW/pmc1 -o pmc1 -- $CMD > $RAW/pmc1.log 2>&1
python tools/prof_summarize.py $RAW $OUT/prof_pmc.txt > /dev/null
python - <<'PY'
import re
rows = [l for l in open("gpurun_out/prof_pmc.txt").read().split("# counters")[1].splitlines()[1:] if l.strip()]
hdr = rows[0].split()
print("kernel".ljust(28), "calls  us/call  mfma_util  wait_any%  wait_inst%  active%  waves  cyc/wave(k)")
import csv
for l in rows[1:40]:
    parts = l.split()
    vals = list(map(float, parts[-8:])); name = " ".join(parts[:-9]); calls = parts[-9]
    gui, active, busy, mfma, wait_any, wait_inst, waves, wcyc = vals  # alphabetical: GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES
    if mfma == 0: continue
    util = (mfma / 1024) / (gui / 8)
    print(name[:28].ljust(28), calls.rjust(5), f"{gui/8/2.3e3:8.0f}", f"{util:9.2f}", f"{100*wait_any/wcyc:9.1f}", f"{100*wait_inst/wcyc:10.1f}", f"{100*active/wcyc:8.1f}", f"{waves:7.0f}", f"{4*wcyc/waves/1e3:9.1f}")
PY

#!/bin/bash
export TMPDIR=/tmp
RAW=/tmp/prof_gen
rm -rf $RAW; mkdir -p $RAW gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python tools/gpu_gen_probe.py ${1:-10000} > $RAW/trace.log 2>&1
grep "^{" $RAW/trace.log
python tools/prof_summarize.py $RAW gpurun_out/gen_summary.txt | head -40

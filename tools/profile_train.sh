#!/bin/bash
# rocprofv3 kernel trace of the training-step pieces (tools/gpu_train_bench.py); reduced table only.
export TMPDIR=/tmp
RAW=/tmp/prof_raw_train
OUT=$PWD/gpurun_out
rm -rf $RAW; mkdir -p $RAW $OUT
CMD="python tools/gpu_train_bench.py ${1:-2}"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $RAW/trace.log 2>&1
tail -25 $RAW/trace.log
python tools/prof_summarize.py $RAW $OUT/prof_train_summary.txt | head -50

#!/bin/bash
# rocprofv3 kernel trace of forward + second-order pass (4 repetitions); reduced table only
export TMPDIR=/tmp
RAW=/tmp/prof_t2; rm -rf $RAW; mkdir -p $RAW gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python tools/debug/train2_loop.py ${1:-2} > $RAW/trace.log 2>&1
python tools/prof_summarize.py $RAW gpurun_out/train2_trace.txt | head -${ROWS:-60}

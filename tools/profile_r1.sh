#!/bin/bash
# rocprofv3 evidence for one bench configuration; only the reduced tables are kept.
export TMPDIR=/tmp
RAW=/tmp/prof_raw
OUT=$PWD/gpurun_out
rm -rf $RAW; mkdir -p $RAW $OUT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- $CMD > $RAW/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d $RAW/pmc1 -o pmc1 -- $CMD > $RAW/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $RAW/pmc2 -o pmc2 -- $CMD > $RAW/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $RAW/pmc3 -o pmc3 -- $CMD > $RAW/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVES -d $RAW/pmc4 -o pmc4 -- $CMD > $RAW/pmc4.log 2>&1
find $RAW -type f | head -30
tail -3 $RAW/trace.log
python tools/prof_summarize.py $RAW $OUT/prof_summary.txt | head -80
cp $(find $RAW -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null

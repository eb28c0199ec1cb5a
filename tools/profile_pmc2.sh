#!/bin/bash
# second SQ counter pass (instruction mix, LDS) on the bench; per-kernel averages printed and saved
export TMPDIR=/tmp
RAW=/tmp/prof_raw2
OUT=$PWD/gpurun_out
rm -rf $RAW; mkdir -p $RAW $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $RAW/pmc1 -o pmc1 -- $CMD > $RAW/pmc1.log 2>&1
tail -3 $RAW/pmc1.log
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR -d $RAW/pmc2 -o pmc2 -- $CMD > $RAW/pmc2.log 2>&1
tail -3 $RAW/pmc2.log
python tools/prof_summarize.py $RAW $OUT/prof_pmc2.txt > /dev/null
python - <<'PY'
txt = open("gpurun_out/prof_pmc2.txt").read()
for sec in txt.split("# counters")[1:]:
    lines = [l for l in sec.splitlines() if l.strip()]
    print(lines[1][:300])
    for l in lines[2:10]:
        print(l[:300])
PY

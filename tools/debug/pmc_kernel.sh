#!/bin/bash
# SQ counter passes of a command, reduced to per-kernel per-wave figures for kernels matching a pattern:
#   bash tools/debug/pmc_kernel.sh <pattern> <command ...>
export TMPDIR=/tmp
PAT=$1; shift
RAW=/tmp/pmc_raw; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $RAW/p1 -o p1 -- "$@" > $RAW/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $RAW/p2 -o p2 -- "$@" > $RAW/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES -d $RAW/p3 -o p3 -- "$@" > $RAW/p3.log 2>&1
PAT=$PAT python3 - <<'PY'
import csv, glob, os, collections
pat = os.environ["PAT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for f in glob.glob("/tmp/pmc_raw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        k = k.split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, f)].add(r["Dispatch_Id"])
for k, c in agg.items():
    n = max(len(v) for (kk, f), v in calls.items() if kk == k)
    waves = c.get("SQ_WAVES", 0) / n or 1
    print(f"== {k}: {n} dispatches, {waves:.0f} waves each")
    for name in sorted(c):
        v = c[name] / n
        print(f"   {name:32s} {v:16.0f}   per wave {v / waves:12.1f}")
PY

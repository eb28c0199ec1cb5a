#!/bin/bash
# A/B of an environment switch on bench_soap.py, alternating on one box:  bash tools/debug/ab_soap_env.sh VAR v0 v1 [reps]
VAR=$1; A=$2; B=$3; REPS=${4:-2}
for i in $(seq $REPS); do
  for v in $A $B; do
    env $VAR=$v python bench_soap.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']), round(d['ms_per_step'],3), d['roofline']['stages_ms'])"
  done
done

#!/bin/bash
# A/B of one pet_config_set switch on the headline bench, alternating on the same box:  bash tools/debug/ab_switch.sh key v0 v1 [reps]
KEY=$1; A=$2; B=$3; REPS=${4:-2}
for i in $(seq $REPS); do
  for v in $A $B; do
    python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --set $KEY=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$KEY=$v', round(d['value']), round(d['ms_per_step'],3))"
  done
done

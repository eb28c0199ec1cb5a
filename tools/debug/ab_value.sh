#!/bin/bash
# usage: ab_value.sh "k1=v1 k2=v2" ... : the bench value (20 steps) for each set of pet_config_set switches, twice
for cfg in "$@"; do
  sets=""; for kv in $cfg; do sets="$sets --set $kv"; done
  for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline $sets 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'value', round(d['value']), 'ms', round(d['ms_per_step'],3))"
  done
done

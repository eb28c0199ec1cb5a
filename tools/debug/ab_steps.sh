#!/bin/bash
# usage: ab_steps.sh KEY V1 V2 ...: the production (two-stream) step time for each value, 3 repetitions of 20 steps
key=$1; shift
for rep in 1 2 3; do
for v in "$@"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --set $key=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$key=$v', round(d['ms_per_step'],3), round(d['value']))"
done; done

#!/bin/bash
# usage: ab.sh KEY V1 V2 ... : stage table rows + step time for each value of a pet_config_set switch
key=$1; shift
for v in "$@"; do
  echo "== $key=$v"
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-all --set $key=$v 2>&1 | grep -v "^{" | grep "ms x" | head -${ROWS:-12}
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --set $key=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], 'E', d['config']['total_energy_rank0'])"
done

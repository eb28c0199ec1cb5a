#!/bin/bash
# usage: ab_multi.sh "k1=v1 k2=v2" "k1=v3" ... : stage table + step time for each set of pet_config_set switches
for cfg in "$@"; do
  sets=""; for kv in $cfg; do sets="$sets --set $kv"; done
  echo "== $cfg"
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-all $sets 2>&1 | grep -v "^{" | grep "ms x" | head -${ROWS:-10}
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline $sets 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], 'E', d['config']['total_energy_rank0'])"
done

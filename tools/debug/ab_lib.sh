#!/bin/bash
# A/B of variant libraries (tools/prof_lib/<v>/libpet_hip.so) against the product library:
#   bash tools/debug/ab_lib.sh <variant...>   -> per-stage ms (stage_times.py) and ms per step (bench.py --no-extras)
cp metatrain_amd/lib/libpet_hip.so /tmp/lib_base.so
for v in base "$@" base "$@"; do
  if [ $v = base ]; then cp /tmp/lib_base.so metatrain_amd/lib/libpet_hip.so; else cp tools/prof_lib/$v/libpet_hip.so metatrain_amd/lib/libpet_hip.so; fi
  TAG=$v python tools/debug/stage_times.py 2>/dev/null | tail -1
  python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   $v ms_per_step', round(d['ms_per_step'],3), 'E=', d['config'].get('total_energy_rank0'))
"
done
cp /tmp/lib_base.so metatrain_amd/lib/libpet_hip.so

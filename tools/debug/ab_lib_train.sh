#!/bin/bash
# A/B of variant libraries on the training step:  bash tools/debug/ab_lib_train.sh <variant...>
cp metatrain_amd/lib/libpet_hip.so /tmp/lib_base.so
for v in base "$@" base "$@"; do
  if [ $v = base ]; then cp /tmp/lib_base.so metatrain_amd/lib/libpet_hip.so; else cp tools/prof_lib/$v/libpet_hip.so metatrain_amd/lib/libpet_hip.so; fi
  python bench_train.py --steps 5 --warmup 2 --no-two-micro 2>/dev/null | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
st=d['roofline'].get('stages_ms') or {}
print('$v ms_per_step', round(d['ms_per_step'],2), 'loss', d['config'].get('loss_first_last'), {k: round(v,2) for k,v in list(st.items())[:10]})
"
done
cp /tmp/lib_base.so metatrain_amd/lib/libpet_hip.so

cp metatrain_amd/lib/libpet_hip.so /tmp/lib_base.so
for v in base $1 base $1 base $1 base $1; do
  if [ $v = base ]; then cp /tmp/lib_base.so metatrain_amd/lib/libpet_hip.so; else cp tools/prof_lib/$v/libpet_hip.so metatrain_amd/lib/libpet_hip.so; fi
  timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline $AB_ARGS 2>/dev/null | python3 -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   $v ms_per_step', round(d['ms_per_step'],3))
"
done
cp /tmp/lib_base.so metatrain_amd/lib/libpet_hip.so

#!/bin/bash
# A/B of variant libraries (tools/prof_lib/<v>/libpet_hip.so) against the product library on the SOAP-BPNN bench:
#   bash tools/debug/ab_soap_lib.sh <variant...>   (extra bench_soap.py arguments in SOAP_ARGS, e.g. "--set soap_packed=0")
cp metatrain_amd/lib/libpet_hip.so /tmp/lib_base.so
for v in base "$@" base "$@"; do
  if [ $v = base ]; then cp /tmp/lib_base.so metatrain_amd/lib/libpet_hip.so; else cp tools/prof_lib/$v/libpet_hip.so metatrain_amd/lib/libpet_hip.so; fi
  python bench_soap.py --no-cpu-baseline $SOAP_ARGS 2>/dev/null | tail -1 | python3 -c "
import sys, json
d=json.loads(sys.stdin.read())
print('$v $SOAP_ARGS ms_per_step', round(d['ms_per_step'],3), d['roofline']['stages_ms'], 'E=', d['config'].get('total_energy_rank0'))
"
done
cp /tmp/lib_base.so metatrain_amd/lib/libpet_hip.so

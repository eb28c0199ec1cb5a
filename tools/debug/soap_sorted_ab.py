"""soap_sorted = 1 (species-sorted tail tiles: gathered rows) against 0 (all networks stacked: rows in atom order) and the
alchemical model (one network, rows in atom order): where does the feature-streaming tail lose its bandwidth?"""
import subprocess, sys, os
for mode, extra in (("sorted=1", []), ("sorted=0", []), ("alchemical", ["alchemical"])):
    code = f"""
import sys; sys.argv=['x','100000']+{extra!r}
sys.path.insert(0,'.')
from metatrain_amd import runtime as rt
rt.config_set('soap_sorted', {0 if mode=='sorted=0' else 1})
exec(open('tools/gpu_soap_bench.py').read())
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout
    print("==", mode); print("\n".join(out.strip().splitlines()[-9:]))

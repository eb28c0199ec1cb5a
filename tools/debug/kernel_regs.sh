#!/bin/bash
# registers / scratch / LDS of the kernels in the built library (device code object metadata):
#   bash tools/debug/kernel_regs.sh k_emlp_bwd_p2 k_comb_bwd_p2
LIB=${LIB:-metatrain_amd/lib/libpet_hip.so}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=<(objcopy -O binary --only-section=.hip_fatbin $LIB /dev/stdout) --output=$TMP/dev.co --unbundle 2>/dev/null \
  || { objcopy -O binary --only-section=.hip_fatbin $LIB $TMP/fat.bin; /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$TMP/fat.bin --output=$TMP/dev.co --unbundle; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co > $TMP/notes.txt
python3 - $TMP/notes.txt "$@" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for blk in txt.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or (pats and not any(p in name.group(1) for p in pats)):
        continue
    g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [None, "?"])[1]
    agpr = blk.split("\n")[0].strip()
    print(f"{name.group(1)[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {agpr:>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s} spill {g('vgpr_spill_count'):>3s}")
PY
rm -rf $TMP

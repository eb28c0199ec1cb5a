#!/bin/bash
# registers / scratch / LDS of the kernels in the built library (every code object of the fat binary: the translation units
# compiled without -fgpu-rdc carry their own):  bash tools/debug/kernel_regs.sh k_emlp_bwd_p2 k_comb_bwd_p2
LIB=${LIB:-metatrain_amd/lib/libpet_hip.so}
TMP=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $LIB $TMP/fat.bin
python3 - $TMP/fat.bin $TMP <<'PY'
import sys
data=open(sys.argv[1],'rb').read(); out=sys.argv[2]
# split concatenated clang offload bundles
magic=b"__CLANG_OFFLOAD_BUNDLE__"
idx=[]; i=data.find(magic)
while i>=0: idx.append(i); i=data.find(magic,i+1)
idx.append(len(data))
for n,(a,b) in enumerate(zip(idx[:-1],idx[1:])): open(f"{out}/b{n}.bin","wb").write(data[a:b])
print(len(idx)-1)
PY
for f in $TMP/b*.bin; do /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$f --output=$f.co --unbundle 2>/dev/null && /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f.co >> $TMP/notes.txt; done
python3 - $TMP/notes.txt "$@" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for blk in txt.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name or (pats and not any(p in name.group(1) for p in pats)): continue
    g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [None, "?"])[1]
    print(f"{name.group(1)[:48]:48s} vgpr {g('vgpr_count'):>4s} agpr {blk.split(chr(10))[0].strip():>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s} spill {g('vgpr_spill_count'):>3s}")
PY
rm -rf $TMP

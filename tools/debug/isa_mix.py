"""Instruction mix of one kernel in a -save-temps .s file, per basic block:  python tools/debug/isa_mix.py file.s <mangled-name substring>"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sys.argv[2] in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("v_cvt"): return "cvt"
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")): return "trans"
    if op.startswith(("v_mov", "v_perm", "v_bfi", "v_and", "v_or", "v_lshl", "v_lshr", "v_cndmask", "v_bfe")): return "v_move/bit"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "ds_bpermute", "ds_swizzle")): return "xlane"
    if op.startswith("v_"): return "v_arith"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
blocks, cur, name = [], collections.Counter(), "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append((name, cur)); cur, name = collections.Counter(), t.split(":")[0]
        continue
    m = re.match(r"^([a-z_0-9]+)\s", t + " ")
    if not m or t.startswith((";", ".")): continue
    cur[kind(m.group(1))] += 1
blocks.append((name, cur))
tot = collections.Counter()
for n, c in blocks: tot.update(c)
keys = ["mfma", "acc_mov", "cvt", "v_pk", "v_arith", "v_move/bit", "trans", "xlane", "lds", "vmem", "scratch", "salu", "wait", "nop"]
print("block".ljust(12), " ".join(k.rjust(10) for k in keys))
for n, c in blocks:
    if sum(c.values()) >= 150: print(n.ljust(12), " ".join(str(c[k]).rjust(10) for k in keys))
print("TOTAL".ljust(12), " ".join(str(tot[k]).rjust(10) for k in keys))

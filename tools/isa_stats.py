#!/usr/bin/env python3
"""Instruction mix of one kernel from hipcc -S output, per basic block (largest first).
usage: tools/isa_stats.py file.s <mangled-name-substring> [n_blocks]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 6
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple([":"] + [")" ])) is False or (l.startswith("_Z") and key in l.split(":")[0] and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))


def kind(op):
    if op.startswith(("v_writelane", "v_readlane", "v_readfirstlane")):
        return "lane<->sgpr"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "v_cmp"
    if op.startswith("v_") and "_f64" in op:
        return "valu_f64"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"):
        return "vmem_store"
    if op.startswith("global_atomic"):
        return "vmem_atomic"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


blocks, cur, name = [], collections.Counter(), "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", ".")) and not re.match(r"^\.LBB\S+:", t):
        continue
    m = re.match(r"^(\.LBB\S+):", t)
    if m:
        blocks.append((name, cur))
        cur, name = collections.Counter(), m.group(1)
        continue
    cur[kind(t.split()[0])] += 1
blocks.append((name, cur))
tot = collections.Counter()
for _, c in blocks:
    tot.update(c)
print("kernel lines %d..%d, %d blocks; whole function:" % (start, end, len(blocks)), dict(tot))
for name, c in sorted(blocks, key=lambda b: -sum(b[1].values()))[:nblk]:
    print("%-14s %5d  %s" % (name, sum(c.values()), dict(sorted(c.items()))))

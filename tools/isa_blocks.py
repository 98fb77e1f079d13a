#!/usr/bin/env python3
"""Instruction census of one kernel in a hipcc -S listing, per basic block: VALU / conversions / SALU / LDS / global / waits.
Usage: tools/isa_blocks.py file.s <substring of the mangled kernel name> [--dump LABEL]"""
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    fn, key = sys.argv[1], sys.argv[2]
    dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
    lines = open(fn).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") or s.startswith(".Lfunc_end"):
            break
        s = s.split(";")[0].strip()
        if not s or s.startswith("."):
            if s.endswith(":") and s.startswith(".LBB"):
                cur = s[:-1]
                blocks[cur] = []
            continue
        if s.endswith(":"):
            continue
        blocks[cur].append(s)
    tot = Counter()
    for name, ins in blocks.items():
        c = Counter(classify(x.split()[0]) for x in ins)
        tot.update(c)
        tgt = [x.split()[-1] for x in ins if x.startswith(("s_cbranch", "s_branch"))]
        print("%-12s n=%4d  valu %4d cvt %3d salu %4d lds %3d vmem %3d wait %3d bar %d  -> %s" % (
            name, len(ins), c["valu"], c["cvt"], c["salu"], c["lds"], c["vmem"], c["wait"], c["barrier"], ",".join(tgt)))
        if dump and name == dump:
            print("\n".join("    " + x for x in ins))
    print("total", dict(tot))


if __name__ == "__main__":
    main()

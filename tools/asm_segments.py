#!/usr/bin/env python3
"""Instruction mix of one kernel between its s_barrier instructions (from /tmp/amenv.s, written by tools/asm_stats.py).
Usage: python tools/asm_segments.py <mangled-name-prefix>"""
import collections
import sys

lines = open("/tmp/amenv.s").read().split("\n")
i0 = [i for i, l in enumerate(lines) if l.startswith(sys.argv[1])][0]
i1 = next(i for i in range(i0, len(lines)) if ".end_amdhsa_kernel" in lines[i])
body = [l.strip() for l in lines[i0:i1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]


def cls(op):
    for pre, name in (("v_mfma", "mfma"), ("v_accvgpr", "acc"), ("v_readlane", "lane"), ("v_writelane", "lane"), ("v_", "valu"), ("ds_", "ds"), ("global_load", "gload"),
                      ("global_store", "gstore"), ("scratch", "scratch"), ("s_waitcnt", "wait"), ("s_nop", "nop"), ("s_cbranch", "branch"), ("s_", "salu")):
        if op.startswith(pre):
            return name
    return "other"


seg, cur = [], collections.Counter()
for t in body:
    op = t.split()[0]
    if op == "s_barrier":
        seg.append(cur)
        cur = collections.Counter()
    cur[cls(op)] += 1
    cur["n"] += 1
seg.append(cur)
for k, c in enumerate(seg):
    print(k, dict(sorted(c.items())))

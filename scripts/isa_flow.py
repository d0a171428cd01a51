#!/usr/bin/env python3
"""Compressed instruction flow of one kernel out of /tmp/co/k.s (scripts/kernel_resources.py --disasm): memory / LDS / MFMA / barrier / wait
instructions with run lengths — shows at a glance whether LDS reads travel ahead of the matrix products or sit in front of a wait each.
    python scripts/isa_flow.py <mangled-name-substring> [first] [last]"""
import re, sys
name = sys.argv[1]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
keep = re.compile(r"v_mfma|s_barrier|s_waitcnt|ds_read|ds_write|ds_bpermute|global_load|global_store|scratch_|buffer_|s_cbranch|s_endpgm")
inside = False
out = []
with open("/tmp/co/k.s") as f:
    for line in f:
        if not inside:
            if line.rstrip().endswith(">:") and name in line:
                inside = True
            continue
        if line.rstrip().endswith(">:") and "<" in line and not line.startswith("\t") and name not in line:
            break
        t = line.split()
        if not t:
            continue
        op = t[0]
        if keep.search(op):
            out.append(op + (" " + t[1] if op == "s_waitcnt" else ""))
        if op == "s_endpgm":
            break
prev, n, k = None, 0, 0
for o in out + [None]:
    if o == prev:
        n += 1
        continue
    if prev is not None:
        if lo <= k < hi:
            print(f"{n:4d} {prev}")
        k += 1
    prev, n = o, 1

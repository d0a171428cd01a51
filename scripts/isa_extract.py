#!/usr/bin/env python3
"""One kernel's instructions out of /tmp/co/k.s (scripts/kernel_resources.py --disasm) into a file: isa_extract.py <mangled-substring> <out.s>"""
import sys
name, out = sys.argv[1], sys.argv[2]
lines, inside = [], False
for line in open("/tmp/co/k.s"):
    if not inside:
        if line.rstrip().endswith(">:") and name in line:
            inside = True
        continue
    t = line.split()
    if not t:
        continue
    lines.append(line.split("//")[0].rstrip())
    if t[0] == "s_endpgm":
        break
open(out, "w").write("\n".join(lines) + "\n")
print(len(lines), "instructions")

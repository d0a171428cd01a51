#!/usr/bin/env python3
"""Register / scratch usage of the kernels in ONE object file of the library (rxinfer.jl_amd/csrc/obj/*.o), without linking:
    python scripts/obj_resources.py rxinfer.jl_amd/csrc/obj/dense_nt4.o [pattern]"""
import os, re, subprocess, sys, tempfile

L = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as td:
    import glob, shutil
    shutil.copy(obj, os.path.join(td, "o.o"))
    subprocess.check_call([f"{L}/llvm-objdump", "--offloading", "o.o"], cwd=td, stdout=subprocess.DEVNULL)
    txt = "".join(subprocess.check_output([f"{L}/llvm-readelf", "--notes", co], text=True) for co in sorted(glob.glob(os.path.join(td, "o.o.*gfx950*"))))
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s}")

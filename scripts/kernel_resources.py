#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of the built library, read from the metadata of its code objects (one per translation
unit: rxinfer.jl_amd/csrc/launch_tables.hpp).
    python scripts/kernel_resources.py [pattern] [--disasm]     (leaves /tmp/co/*.co and, with --disasm, /tmp/co/k.s)"""
import glob, os, re, shutil, subprocess, sys

L = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("SO", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd", "csrc", "librxhip.so"))
shutil.rmtree("/tmp/co", ignore_errors=True)
os.makedirs("/tmp/co", exist_ok=True)
shutil.copy(so, "/tmp/co/lib.so")
subprocess.check_call([f"{L}/llvm-objdump", "--offloading", "lib.so"], cwd="/tmp/co", stdout=subprocess.DEVNULL)
cos = sorted(glob.glob("/tmp/co/lib.so.*gfx950*"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
pat = args[0] if args else ""
for i, co in enumerate(cos):
    os.rename(co, f"/tmp/co/k{i}.co")
    txt = subprocess.check_output([f"{L}/llvm-readelf", "--notes", f"/tmp/co/k{i}.co"], text=True)
    for blk in txt.split("- .agpr_count:")[1:]:
        def g(k):
            m = re.search(r"\." + k + r":\s*(\S+)", blk)
            return m.group(1) if m else "?"
        name = g("name")
        try:
            name = subprocess.check_output([f"{L}/llvm-cxxfilt", name], text=True).strip()
        except Exception:
            pass
        if pat in name:
            print(f"{name[:100]:100s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>4s} "
                  f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s} unit {i}")
if "--disasm" in sys.argv:
    with open("/tmp/co/k.s", "w") as f:
        for i in range(len(cos)):
            subprocess.check_call([f"{L}/llvm-objdump", "-d", f"/tmp/co/k{i}.co"], stdout=f)

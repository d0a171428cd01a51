#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of the built library, read from the code object's metadata.
    python scripts/kernel_resources.py [pattern] [--disasm]     (leaves /tmp/co/k.co and, with --disasm, /tmp/co/k.s)"""
import os, re, subprocess, sys

L = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("SO", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd", "csrc", "librxhip.so"))
os.makedirs("/tmp/co", exist_ok=True)
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, "/tmp/co/fat.bin"])
subprocess.check_call([f"{L}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                       "--input=/tmp/co/fat.bin", "--output=/tmp/co/k.co", "--unbundle"])
txt = subprocess.check_output([f"{L}/llvm-readelf", "--notes", "/tmp/co/k.co"], text=True)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
pat = args[0] if args else ""
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
              f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
if "--disasm" in sys.argv:
    with open("/tmp/co/k.s", "w") as f:
        subprocess.check_call([f"{L}/llvm-objdump", "-d", "/tmp/co/k.co"], stdout=f)

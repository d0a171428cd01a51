"""The Julia `ccall` layer mirrors the C structs field by field (no Julia toolchain exists in the image, so nothing else would
notice a descriptor that grew on one side only): field names and order of every mirrored struct, header against RxHip.jl."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "rxhip.h")).read()
JULIA = open(os.path.join(ROOT, "rxinfer.jl_amd", "julia", "RxHip.jl")).read()

PAIRS = {"rxhip_lgssm_desc": "LgssmDesc", "rxhip_graph_desc": "GraphDesc", "rxhip_lgssm_lowered": "LgssmLowered",
         "rxhip_gmm_desc": "GmmDesc", "rxhip_mvgmm_desc": "MvGmmDesc", "rxhip_hgf_desc": "HgfDesc",
         "rxhip_drift_chain_desc": "DriftChainDesc"}


def c_fields(name):
    end = re.search(r"\}\s*" + name + r"\s*;", HEADER).start()
    start = HEADER.rfind("typedef struct", 0, end)
    body = HEADER[HEADER.index("{", start) + 1:end]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(unsigned\s+)?[A-Za-z_0-9]+\s*", "", decl, count=1)   # drop the type
        out += [re.sub(r"[\*\s]|\[.*?\]", "", v) for v in decl.split(",")]
    return [f for f in out if f]


def julia_fields(name):
    body = re.search(r"(?:mutable\s+)?struct\s+" + name + r"\b(.*?)\nend", JULIA, re.S).group(1)
    out = []
    for line in body.splitlines():
        line = line.split("#")[0]
        if "=" in line and "new(" in line:
            continue
        out += re.findall(r"([A-Za-z_0-9]+)::", line)
    return out


def test_every_mirrored_struct_has_the_header_fields_in_order():
    for cname, jname in PAIRS.items():
        assert julia_fields(jname) == c_fields(cname), (cname, jname)

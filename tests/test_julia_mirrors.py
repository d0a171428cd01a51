"""The Julia `ccall` layer mirrors the C structs field by field (no Julia toolchain exists in the image, so nothing else would
notice a descriptor that grew on one side only): field names and order of every mirrored struct, header against RxHip.jl."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "rxhip.h")).read()
JULIA = open(os.path.join(ROOT, "rxinfer.jl_amd", "julia", "RxHip.jl")).read()

PAIRS = {"rxhip_lgssm_desc": "LgssmDesc", "rxhip_graph_desc": "GraphDesc", "rxhip_lgssm_lowered": "LgssmLowered",
         "rxhip_gmm_desc": "GmmDesc", "rxhip_mvgmm_desc": "MvGmmDesc", "rxhip_hgf_desc": "HgfDesc",
         "rxhip_drift_chain_desc": "DriftChainDesc", "rxhip_noise_prior": "NoisePrior",
         "rxhip_lgssm_noise_lowered": "LgssmNoiseLowered", "rxhip_lgssm_lowered ": "LgssmLoweredFields",
         "rxhip_tree_info": "TreeInfo", "rxhip_rule_call": "RuleCall"}


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
        assert julia_fields(jname) == c_fields(cname.strip()), (cname, jname)   # (a trailing blank: a second mirror of the same C struct)


PLUGIN = open(os.path.join(ROOT, "rxinfer.jl_amd", "julia", "HIPInferencePlugin.jl")).read()


def test_plugin_holds_no_second_copy_of_a_c_struct():
    """A C descriptor is mirrored ONCE, in RxHip.jl (checked above).  The plugin once carried a private copy of rxhip_graph_desc
    that nothing used and that had already lost a field: any struct of the plugin whose fields are pointers is
    such a copy."""
    for m in re.finditer(r"(?:mutable\s+)?struct\s+([A-Za-z_0-9]+)[^\n]*\n(.*?)\nend", PLUGIN, re.S):
        assert "::Ptr{" not in m.group(2), f"struct {m.group(1)} of the plugin mirrors a C struct: keep the mirror in RxHip.jl"


def test_node_vocabulary_of_the_plugin_is_the_header_enum():
    """hip_node(fform) -> (code, name, interfaces) against RXHIP_NODE_* and the interface orders the header documents."""
    enum = {int(v): k for k, v in re.findall(r"RXHIP_NODE_([A-Z_]+)\s*=\s*(\d+)", HEADER)}
    codes = {}
    for code, name, ifaces in re.findall(r"hip_node\(.*?\)\s*=\s*\(Int32\((\d+)\),\s*\"([^\"]+)\",\s*\(([^)]*)\)\)", PLUGIN):
        codes[int(code)] = (name, [x.strip().lstrip(":") for x in ifaces.split(",") if x.strip()])
    assert sorted(codes) == sorted(enum), (sorted(codes), sorted(enum))     # every node of the header, no other
    want = {"MVNORMAL_MEAN_COV": "MvNormalMeanCovariance", "MULTIPLY": "*", "NORMAL_MEAN_VARIANCE": "NormalMeanVariance",
            "NORMAL_MEAN_PRECISION": "NormalMeanPrecision", "GAMMA_SHAPE_RATE": "GammaShapeRate", "DIRICHLET": "Dirichlet",
            "BETA": "Beta", "CATEGORICAL": "Categorical", "BERNOULLI": "Bernoulli", "NORMAL_MIXTURE": "NormalMixture", "GCV": "GCV",
            "WISHART": "Wishart", "ADD": "+", "MVNORMAL_MEAN_PRECISION": "MvNormalMeanPrecision", "GAMMA_SHAPE_SCALE": "GammaShapeScale"}
    for code, cname in enum.items():
        assert codes[code][0] == want[cname], (code, cname, codes[code])
    # interface order: the tuple in the header comment of each enumerator, where it gives one
    for cname, tup in re.findall(r"RXHIP_NODE_([A-Z_]+)\s*=\s*\d+,?\s*/\*\s*\(([^)]*)\)", HEADER):
        code = next(k for k, v in enum.items() if v == cname)
        hdr = [re.sub(r"\[.*?\]", "", x).strip() for x in tup.split(",")]
        assert codes[code][1] == hdr, (cname, codes[code][1], hdr)
    # the Python mirror's constants are the same enum
    import sys
    sys.path.insert(0, os.path.join(ROOT, "rxinfer.jl_amd"))
    from rxhip import _lib
    for code, cname in enum.items():
        assert getattr(_lib, "NODE_" + cname) == code, cname

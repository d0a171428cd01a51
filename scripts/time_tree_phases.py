"""The executor's two phases (sweep; Bethe terms) under the resident-levels and the walk schedule: device ms per iteration with and without the free energy."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
os.environ["RXHIP_TEST_HOOKS"] = "1"
from rxhip import workloads  # noqa: E402
from rxhip.graph import two_branch_chain_graph  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

mdl = workloads.c1_model()
T = 128
gb, xs, ys = two_branch_chain_graph(T, mdl["A"], mdl["B"][:2], mdl["B"][2:], mdl["P"], mdl["Q"][:2, :2], mdl["Q"][2:, 2:], mdl["m0"], mdl["V0"])
for R in (8192, 16384, 32768, 65536, 131072):
    rows = np.random.default_rng(0).standard_normal((R, T * 4)) * 3.0
    for mode in (1, 2):
        os.environ["RXHIP_TREE_MODE"] = str(mode)
        with TreeEngine(gb, n_replicas=R) as eng:
            eng.set_data(ys, rows)
            out = []
            for fe in (False, True):
                eng.run(1, fe)
                best = 1e9
                for _ in range(3):
                    eng.run(1, fe)
                    best = min(best, eng.last_iteration_ms())
                out.append(best)
            print(f"R={R:6d} mode={mode} rb={eng.info['replicas_per_workgroup']:3d}: sweep {out[0]:7.3f} ms, with free energy {out[1]:7.3f} ms, Bethe phase {out[1] - out[0]:6.3f} ms", flush=True)

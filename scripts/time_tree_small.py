"""The executor at small batches: a launch per level (mode 0) against workgroup-resident levels (mode 1), a deep chain and a wide star."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["RXHIP_TEST_HOOKS"] = "1"
import tree_graphs as tg  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

graphs = [("two_branch T=128 d=4", tg.two_branch_chain(T=128, d=4, dy1=2, dy2=2)), ("star 3000 leaves d=3", tg.star(n_leaves=3000, d=3)),
          ("branching tree depth 7 d=2", tg.branching_tree(depth=7, fanout=2, d=2))]
for name, (gb, ys, _) in graphs:
    for R in (1, 16, 64, 256, 1024, 4096):
        data = tg.random_data(gb, ys, R, 0)
        res = []
        for mode in (None, 0, 1):
            if mode is None:
                os.environ.pop("RXHIP_TREE_MODE", None)
            else:
                os.environ["RXHIP_TREE_MODE"] = str(mode)
            with TreeEngine(gb, n_replicas=R) as eng:
                eng.set_data(ys, data)
                eng.run(1, True)
                best = 1e9
                for _ in range(3):
                    eng.run(1, True)
                    best = min(best, eng.last_iteration_ms())
                res.append((eng.info["mode"], best, eng.info["n_levels"], eng.info["n_ops"]))
        print(f"{name:28s} R={R:5d} levels={res[0][2]:4d} ops={res[0][3]:6d}: default mode {res[0][0]} {res[0][1]:8.3f} ms | mode 0 {res[1][1]:8.3f} | mode 1 {res[2][1]:8.3f}", flush=True)

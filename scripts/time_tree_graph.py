"""The launch-per-level schedule replayed as a HIP graph (RXHIP_TREE_GRAPH=1, the default) against direct launches (=0): small batches, where an iteration is
hundreds of short launches."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["RXHIP_TEST_HOOKS"] = "1"
import tree_graphs as tg  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

for d, T in ((16, 64), (32, 32), (8, 64), (4, 64), (64, 16)):
    for R in (1, 16, 256, 1024):
        gb, ys, _ = tg.two_branch_chain(T=T, d=d, dy1=d, dy2=max(1, d // 2))
        data = tg.random_data(gb, ys, R, 0)
        res = {}
        for graph in (0, 1):
            os.environ["RXHIP_TREE_GRAPH"] = str(graph)
            os.environ["RXHIP_TREE_MODE"] = "0"
            with TreeEngine(gb, n_replicas=R) as eng:
                eng.set_data(ys, data)
                eng.run(2, True)
                best = 1e9
                for _ in range(5):
                    eng.run(3, True)
                    best = min(best, eng.last_iteration_ms())
                res[graph] = (best, float(np.ravel(eng.free_energy())[-1]), eng.info["n_levels"])
        assert res[0][1] == res[1][1], res
        print(f"d={d:3d} T={T:3d} R={R:5d} levels={res[0][2]:4d}   direct launches {res[0][0]:8.3f} ms   graph {res[1][0]:8.3f} ms", flush=True)

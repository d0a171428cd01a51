"""The LDS-staged executor kernels (dimensions above 8): time per sweep with the free energy for the two schedules, two observation branches per state."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["RXHIP_TEST_HOOKS"] = "1"
import tree_graphs as tg  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

cases = [(16, 64, 1), (16, 64, 256), (16, 64, 4096), (32, 32, 1), (32, 32, 256), (32, 32, 2048), (64, 16, 1), (64, 16, 256), (8, 64, 256), (12, 64, 256)]
for d, T, R in cases:
    gb, ys, _ = tg.two_branch_chain(T=T, d=d, dy1=d, dy2=max(1, d // 2))
    data = tg.random_data(gb, ys, R, 0)
    for mode, mode_fe in ((0, 0), (2, 2), (2, 0)):   # (sweep schedule, second-phase schedule): a launch per level / a work item per replica walks the op list
        os.environ["RXHIP_TREE_MODE"] = str(mode)
        os.environ["RXHIP_TREE_MODE_FE"] = str(mode_fe)
        with TreeEngine(gb, n_replicas=R) as eng:
            eng.set_data(ys, data)
            eng.run(1, True)
            best = 1e9
            for _ in range(3):
                eng.run(1, True)
                best = min(best, eng.last_iteration_ms())
            inf = eng.info
            calls = eng.counters()["rule_calls"]
        print(f"d={d:3d} T={T:3d} R={R:5d} mode={mode} fe={mode_fe}  {best:9.3f} ms/sweep  ops={inf['n_ops']} levels={inf['n_levels']}  {calls / best * 1e3:10.3e} rule-calls/s  "
              f"{inf['bytes_per_sweep'] * R / best * 1e-6:8.1f} GB/s", flush=True)

"""Register-tile executor kernels (csrc/tree_tile_kernels.hpp): (a) where the walk passes a launch per level, (b) dimensions 5 … 8 on the tile kernels
(RXHIP_TREE_TILE=1) against the lane-per-replica register kernels."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
os.environ["RXHIP_TEST_HOOKS"] = "1"
import tree_graphs as tg  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402


def run(d, T, R, env):
    for k in ("RXHIP_TREE_MODE", "RXHIP_TREE_TILE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    gb, ys, _ = tg.two_branch_chain(T=T, d=d, dy1=d, dy2=max(1, d // 2))
    data = tg.random_data(gb, ys, R, 0)
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.set_data(ys, data)
        eng.run(1, True)
        best = 1e9
        for _ in range(3):
            eng.run(1, True)
            best = min(best, eng.last_iteration_ms())
        fe = float(np.ravel(eng.free_energy())[-1])
    return best, fe


which = sys.argv[1] if len(sys.argv) > 1 else "ab"
if "a" in which:
    for d, T in ((16, 64), (32, 32), (12, 64)):
        for R in (512, 1024, 2048, 8192):
            if d == 32 and R > 4096:
                continue
            t0, _ = run(d, T, R, {"RXHIP_TREE_MODE": "0"})
            t2, _ = run(d, T, R, {"RXHIP_TREE_MODE": "2"})
            print(f"d={d:3d} T={T:3d} R={R:6d}  per level {t0:8.3f} ms   walk {t2:8.3f} ms", flush=True)
if "b" in which:
    for d in (5, 6, 8):
        for R in (1, 256, 4096, 16384, 65536):
            T = 64
            tr, fr = run(d, T, R, {})
            t0, f0 = run(d, T, R, {"RXHIP_TREE_TILE": "1", "RXHIP_TREE_MODE": "0"})
            t2, f2 = run(d, T, R, {"RXHIP_TREE_TILE": "1", "RXHIP_TREE_MODE": "2"})
            print(f"d={d:3d} T={T:3d} R={R:6d}  register kernels {tr:8.3f} ms   tiles per level {t0:8.3f} ms  walk {t2:8.3f} ms   F rel diff {abs(f0 - fr) / abs(fr):.1e} {abs(f2 - fr) / abs(fr):.1e}", flush=True)

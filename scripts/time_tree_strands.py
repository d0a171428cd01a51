"""The executor's sweep under the workgroup-resident (1), walk (2) and strand (3) schedules and its Bethe phase under launch-per-level (0), resident (1) and walk (2):
device ms per iteration, plain benchmark chain and the two-branch chain, d = 4, T = 128."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rxinfer.jl_amd"))
os.environ["RXHIP_TEST_HOOKS"] = "1"
from rxhip import workloads  # noqa: E402
from rxhip.graph import lgssm_graph, two_branch_chain_graph  # noqa: E402
from rxhip.tree import TreeEngine  # noqa: E402

mdl = workloads.c1_model()
T = 128
graphs = {"plain": lgssm_graph(T, mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"]),
          "two_branch": two_branch_chain_graph(T, mdl["A"], mdl["B"][:2], mdl["B"][2:], mdl["P"], mdl["Q"][:2, :2], mdl["Q"][2:, 2:], mdl["m0"], mdl["V0"])}
Rs = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "4096,16384,65536,131072".split(","))]
SWEEP_MODES = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,3".split(","))]
for name, (gb, xs, ys) in graphs.items():
    for R in Rs:
        rows = np.random.default_rng(0).standard_normal((R, T * 4)) * 3.0
        for mode in SWEEP_MODES:
            for mode_fe in (0, 1, 2):
                if mode != 3 and mode_fe != 2:
                    continue
                os.environ["RXHIP_TREE_MODE"] = str(mode)
                os.environ["RXHIP_TREE_MODE_FE"] = str(mode_fe)
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
                    i = eng.info
                    print(f"{name:10s} R={R:6d} sweep mode {mode} fe mode {mode_fe}: sweep {out[0]:7.3f} ms, with free energy {out[1]:7.3f} ms, Bethe phase {out[1] - out[0]:6.3f} ms"
                          f"   [{i['bytes_per_sweep'] * R / out[0] / 1e6:7.1f} GB/s of messages, strands {i['n_strands']} in {i['n_strand_levels']} levels]", flush=True)

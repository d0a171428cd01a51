import os, sys
import numpy as np
sys.path.insert(0, "rxinfer.jl_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
os.environ["RXHIP_TEST_HOOKS"] = "1"; os.environ["RXHIP_TREE_MODE"] = "0"
import tree_graphs as tg, tree_oracle
from rxhip.tree import TreeEngine
for kw, R in ((dict(T=3, d=33, dy1=20, dy2=33), 3), (dict(T=3, d=33, dy1=33, dy2=33), 3), (dict(T=3, d=20, dy1=20, dy2=20), 3), (dict(T=2, d=64, dy1=64, dy2=30), 2)):
    gb, ys, _ = tg.two_branch_chain(**kw)
    data = tg.random_data(gb, ys, R, 0)
    with TreeEngine(gb, n_replicas=R) as eng:
        eng.set_data(ys, data)
        eng.run(1, True)
        g = tree_oracle.TreeGraph(gb.to_dump())
        gv = [v for v in range(len(gb.kind)) if g.gauss[v]]
        post = eng.marginals(gv)
        fe = eng.free_energy_per_replica()
    ref = tree_oracle.infer(gb.to_dump(), tg.data_dict(gb, ys, data[0]))
    worst = []
    for v in gv:
        sd = np.sqrt(np.diag(ref["cov"][v]))
        if np.all(sd < 1e-7):
            continue
        worst.append((float(np.max(np.abs(post[v][0][0] - ref["mean"][v]) / sd)), float(np.max(np.abs(post[v][1][0] - ref["cov"][v]) / np.outer(sd, sd))), v, gb.rows[v]))
    worst.sort(reverse=True)
    print(kw, "worst (mean err, cov err, var, dim):", worst[:4], "fe", fe[0], ref["fe"][-1])

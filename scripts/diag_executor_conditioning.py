import sys, os
sys.path.insert(0,'/root/repo/rxinfer.jl_amd'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, tree_oracle, tree_graphs as tg
from rxhip.graph import two_branch_chain_graph
from rxhip.tree import TreeEngine
rng = np.random.default_rng(0)
for d in (4, 8, 12, 20, 40):
    for v0, qn in ((4.0, 1.0), (1e6, 1e-2), (1e10, 1e-4)):
        q, _ = np.linalg.qr(rng.standard_normal((d, d)))
        h = max(1, d // 2)
        gb, xs, ys = two_branch_chain_graph(6, 0.95 * q, rng.standard_normal((d, d)), rng.standard_normal((h, d)), 0.1 * np.eye(d), qn * np.eye(d), qn * np.eye(h), np.zeros(d), v0 * np.eye(d))
        rows = rng.standard_normal((2, 6 * (d + h)))
        try:
            with TreeEngine(gb, n_replicas=2) as eng:
                eng.set_data(ys, rows); eng.run(1, True)
                post, fe = eng.marginals(xs), eng.free_energy_per_replica()
                kern = eng.info["kernels"]
        except Exception as e:
            print(d, v0, qn, "ERR", str(e)[:100]); continue
        data, o = {}, 0
        for v in ys:
            data[v] = rows[1, o:o + gb.rows[v]]; o += gb.rows[v]
        ref = tree_oracle.infer(gb.to_dump(), data)
        em = max(float(np.max(np.abs(post[v][0][1] - ref["mean"][v]) / np.sqrt(np.diag(ref["cov"][v])))) for v in xs)
        ec = max(float(np.max(np.abs(post[v][1][1] - ref["cov"][v]) / np.outer(np.sqrt(np.diag(ref["cov"][v])), np.sqrt(np.diag(ref["cov"][v]))))) for v in xs)
        print(f"d={d:3d} kernels={kern} V0={v0:8.0e} Q={qn:6.0e}  mean err {em:9.2e} sd   cov rel {ec:9.2e}   fe rel {abs(fe[1]-ref['fe'][0])/abs(ref['fe'][0]):9.2e}")

"""Malformed graph descriptors must come back as a status, never as a crash: valid descriptors of every family the library accepts (state-space chain,
two observation branches per state, shared precision variables, a random forest) with ONE table entry corrupted at a time — kinds, dimensions, pool
offsets, node types, interface indices, the pool itself — through `rxhip_create` (pattern matcher, then the executor's compiler) and `rxhip_tree_create`.
Host logic only: without a GPU a well-formed graph ends in RXHIP_ERR_NO_DEVICE, which is as good as any other status here.  The loop runs in a child
process so that a segmentation fault is a test failure, not the end of the test session."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes, sys
    import numpy as np
    sys.path.insert(0, {root!r} + "/rxinfer.jl_amd"); sys.path.insert(0, {root!r} + "/tests")
    from rxhip import _lib, graph
    import tree_graphs as tg
    L = _lib.lib()
    rng = np.random.default_rng({seed})
    builders = [lambda: tg.two_branch_chain(T=3)[0], lambda: tg.chain_state_noise_precision(T=3, also_obs_noise=True)[0], lambda: tg.random_forest(3, n_steps=6)[0],
                lambda: graph.lgssm_graph(4, np.eye(2) * 0.9, np.eye(2), np.eye(2) * 0.1, np.eye(2), np.zeros(2), np.eye(2))[0], lambda: tg.known_mean_precision(3, 2)[0]]
    seen = {{}}
    for trial in range({trials}):
        gb = builders[trial % len(builders)]()
        g, arrs = gb.tables(n_replicas=int(rng.integers(1, 4)))
        what = rng.integers(0, 9)
        if what == 0: arrs["kind"][rng.integers(0, arrs["kind"].size)] = rng.integers(-2, 6)
        elif what == 1: arrs["rows"][rng.integers(0, arrs["rows"].size)] = rng.choice([-1, 0, 1, 3, 65, 1 << 20])
        elif what == 2: arrs["cols"][rng.integers(0, arrs["cols"].size)] = rng.choice([-1, 0, 2, 7])
        elif what == 3: arrs["coff"][rng.integers(0, arrs["coff"].size)] = rng.choice([-5, -1, 0, arrs["pool"].size - 1, arrs["pool"].size + 10, 1 << 40])
        elif what == 4: arrs["ft"][rng.integers(0, arrs["ft"].size)] = rng.integers(-3, 40)
        elif what == 5: arrs["fi"][rng.integers(0, arrs["fi"].size)] = rng.choice([-1, 0, arrs["kind"].size - 1, arrs["kind"].size, 1 << 33])
        elif what == 6: g.n_const = int(rng.choice([0, 1, max(1, arrs["pool"].size // 2)]))
        elif what == 7: arrs["pool"][rng.integers(0, arrs["pool"].size)] = rng.choice([np.nan, np.inf, -1.0, 0.0, 1e300])
        else: g.n_factors = int(rng.choice([0, -1, max(1, arrs["ft"].size - 1)])) if rng.integers(0, 2) else g.n_factors; g.n_variables = g.n_variables if rng.integers(0, 2) else int(rng.choice([0, -3]))
        for fn in (L.rxhip_create, L.rxhip_tree_create):
            h = ctypes.c_void_p()
            st = fn(ctypes.byref(g), 0, 0, None, ctypes.byref(h)) if fn is L.rxhip_create else fn(ctypes.byref(g), 0, None, ctypes.byref(h))
            seen[st] = seen.get(st, 0) + 1
            if h: L.rxhip_destroy(h)
    print("statuses", dict(sorted(seen.items())))
""")


def test_corrupted_descriptors_are_refused_not_crashed():
    env = dict(os.environ)
    env.pop("RXHIP_TEST_HOOKS", None)
    for seed in (1, 2):
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, seed=seed, trials=400)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
        assert "statuses" in r.stdout

#!/usr/bin/env python3
"""The MFMA chain path (d > 8) as the prior gets vague and the observation noise tight: a random dense model (workloads.random_model), dy = d/2 and dy = d,
V0 = v I, Q scaled by q — posteriors and free energy of `LGSSMEngine` against the oracle's Kalman / RTS restatement, with the condition number of the first
filtered precision V0⁻¹ + Bᵀ Q⁻¹ B beside it.  Run on an MI355X: python scripts/diag_dense_conditioning.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("rxinfer.jl_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import rxhip  # noqa: E402
import rxoracle  # noqa: E402
from rxhip import workloads  # noqa: E402

T, C = 200, 2
for d in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else (8, 16, 32, 64):
    for dy in (max(1, d // 2), d):
        for v in (1e0, 1e2, 1e4, 1e6, 1e8):
            for q in (1e-2, 1.0):
                mdl = workloads.random_model(d, dy, seed=d + dy)
                mdl["V0"] = v * np.eye(d)
                mdl["Q"] = q * mdl["Q"]
                y = workloads.generate_batch(mdl, T, C, seed0=3, threads=1)
                args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
                try:
                    with rxhip.LGSSMEngine(*args, T=T, n_chains=C) as eng:
                        eng.set_data(y)
                        eng.run(1, True)
                        mean, cov = eng.marginals()
                        fe = eng.free_energy_per_chain()
                except Exception as e:
                    print(f"d={d:2d} dy={dy:2d} V0={v:7.0e} Q×{q:5.0e}: {str(e)[:90]}", flush=True)
                    continue
                om, oc, onll = rxoracle.lgssm_kalman_rts(*args, y[:, 0])
                sd = np.sqrt(np.einsum("tii->ti", oc))
                e = max(float(np.max(np.abs(mean[:, 0] - om) / sd)), float(np.max(np.abs(cov[:, 0] - oc) / (sd[:, :, None] * sd[:, None, :]))))
                L1 = np.linalg.inv(mdl["V0"]) + mdl["B"].T @ np.linalg.solve(mdl["Q"], mdl["B"])
                print(f"d={d:2d} dy={dy:2d} V0={v:7.0e} Q×{q:5.0e}: posterior {e:8.2e} sd, free energy {abs(fe[0] - onll) / abs(onll):8.2e}; cond(V0⁻¹ + BᵀQ⁻¹B) = {np.linalg.cond(L1):.1e}", flush=True)

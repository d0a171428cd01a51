#!/usr/bin/env python3
"""Calibration data for the accuracy guard of the MFMA chain path: random dense models with noise scales over decades (tests/fuzz_cases.py run_engine_case's
distribution, d > 8 only), per case the error of `LGSSMEngine` against the oracle's Kalman / RTS restatement and host-computable indicators of how hard the model is.
Prints CSV.  Usage: calib_dense_envelope.py <first seed> <count>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("rxinfer.jl_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import rxhip  # noqa: E402
import rxoracle  # noqa: E402
from rxhip import workloads  # noqa: E402

s0, n = int(sys.argv[1]), int(sys.argv[2])
print("seed,d,dy,T,ptt,err,fe,c_first,c_steady,vague,cV0,cP,cQ,lamQ,stable")
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(seed)
    d = int(rng.choice([9, 12, 16, 17, 24, 32, 33, 48, 64]))
    dy = int(rng.integers(1, d + 1)) if rng.random() < 0.6 else d
    T = int(np.exp(rng.uniform(0.0, np.log(40))))
    ptt = bool(rng.integers(0, 2))
    st = float(rng.uniform(0.3, 0.99))
    mdl = workloads.random_model(d, dy, seed, stable=st)
    mdl["P"] = mdl["P"] * 10.0 ** rng.uniform(-2, 1)
    mdl["Q"] = mdl["Q"] * 10.0 ** rng.uniform(-2, 1)
    mdl["V0"] = mdl["V0"] * 10.0 ** rng.uniform(-1, 3)
    y = workloads.generate_batch(mdl, T, 1, seed0=seed, threads=1)
    args = (mdl["A"], mdl["B"], mdl["P"], mdl["Q"], mdl["m0"], mdl["V0"])
    try:
        with rxhip.LGSSMEngine(*args, T=T, n_chains=1, prior_through_transition=ptt) as eng:
            eng.set_data(y)
            eng.run(1, True)
            mean, cov = eng.marginals()
            fe = eng.free_energy_per_chain()
        om, oc, onll = rxoracle.lgssm_kalman_rts(*args, y[:, 0], prior_through_transition=ptt)
        sd = np.sqrt(np.einsum("tii->ti", oc))
        e = max(float(np.max(np.abs(mean[:, 0] - om) / sd)), float(np.max(np.abs(cov[:, 0] - oc) / (sd[:, :, None] * sd[:, None, :]))))
        ef = abs(fe[0] - onll) / max(1.0, abs(onll))
    except Exception:
        e, ef = float("inf"), float("inf")
    obs = mdl["B"].T @ np.linalg.solve(mdl["Q"], mdl["B"])
    Pi = np.linalg.inv(mdl["P"])
    V0p = mdl["A"] @ mdl["V0"] @ mdl["A"].T + mdl["P"] if ptt else mdl["V0"]
    first = np.linalg.inv(V0p) + obs
    steady = Pi + obs
    ev = np.linalg.eigvalsh
    print(f"{seed},{d},{dy},{T},{int(ptt)},{e:.3e},{ef:.3e},{np.linalg.cond(first):.3e},{np.linalg.cond(steady):.3e},{ev(V0p)[-1] * ev(steady)[-1]:.3e},"
          f"{np.linalg.cond(V0p):.3e},{np.linalg.cond(mdl['P']):.3e},{np.linalg.cond(mdl['Q']):.3e},{ev(obs)[-1]:.3e},{st:.3f}", flush=True)

"""Which of the adversarial fixed-point cases (tests/test_fixed_point_adversarial_gpu.py) deviate from the oracle, by how much, and whether the
full recursion (RXHIP_ELEM_FULL / RXHIP_NO_FROZEN) deviates as well."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("rxinfer.jl_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
os.environ["RXHIP_TEST_HOOKS"] = "1"
import rxhip, rxoracle as rxo
import test_fixed_point_adversarial_gpu as T
rxo.build()

class MP:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=False): os.environ.pop(k, None)

def errs(m, y, mean, cov, fe, chains):
    out = []
    for c in chains:
        om, oc, nll = rxo.lgssm_kalman_rts(m["A"], m["B"], m["P"], m["Q"], m["m0"], m["V0"], np.ascontiguousarray(y[:, c]))
        sd = np.sqrt(np.einsum("tii->ti", oc))
        em = np.abs(mean[:, c] - om) / sd
        ec = np.abs(cov[:, c] - oc) / (sd[:, :, None] * sd[:, None, :])
        out.append((float(em.max()), int(np.unravel_index(em.argmax(), em.shape)[0]), int(np.unravel_index(em.argmax(), em.shape)[1]), float(ec.max()),
                    int(np.unravel_index(ec.argmax(), ec.shape)[0]), float(abs(fe[c] - nll) / abs(nll))))
    return out

cases = []
m4 = T._two_scale_model(4, 0.9999, 6.0, seed=3)
y4 = T._generate(m4, 6000, 4, seed=4)
for pc, C in ((False, 64), (True, 64), (False, 3)):
    cases.append((f"two_scale d4 per_chain={pc} C={C}", m4, np.ascontiguousarray(np.tile(y4, (1, (C + 3) // 4, 1))[:, :C]), dict(per_chain=pc), (0,)))
d = 4
rng = np.random.default_rng(12)
q, _ = np.linalg.qr(rng.standard_normal((d, d)))
mu = dict(A=(1.0 - 1e-6) * (q @ np.diag([1.0, 0.999, 0.99, 0.9]) @ q.T), B=np.eye(d), P=1e-8 * np.eye(d), Q=np.eye(d), m0=np.zeros(d), V0=4.0 * np.eye(d))
yu = np.ascontiguousarray(np.tile(T._generate(mu, 20000, 2, seed=5), (1, 32, 1)))
for pc in (False, True):
    cases.append((f"unit_root d4 per_chain={pc}", mu, yu, dict(per_chain=pc), (0,)))
for dd, TT, seg in ((64, 4000, 0), (48, 3000, 0)):
    mm = T._two_scale_model(dd, 0.9999, 6.0, seed=dd)
    cases.append((f"two_scale d{dd} seg={seg}", mm, T._generate(mm, TT, 1, seed=dd + 1), dict(segments=seg), (0,)))
for name, m, y, kw, chains in cases:
    for full in (False, True):
        t0 = time.time()
        try:
            mean, cov, fe, sched = T._run(m, y, MP(), full=full, **kw)
            print(name, "full" if full else "exits", "sched", sched, "(em, t, comp, ec, t, fe_rel):", errs(m, y, mean, cov, fe, chains), f"{time.time() - t0:.1f}s", flush=True)
        except Exception as e:
            print(name, "full" if full else "exits", "ERROR", e, flush=True)

"""The executor's LDS-staged rule bodies (csrc/tree_wave_kernels.hpp, dimensions above 8) against its register rule bodies (csrc/tree_kernels.hpp), op for op.

Both headers are compiled for the HOST by g++ (tests/host_emul/: a stand-in <hip/hip_runtime.h>, a "wavefront" of one lane) and run over the same hand-made
op tables and random states: what the GPU parity tests established for the register bodies (tests/test_tree_engine_gpu.py against oracle/tree_oracle.py)
carries over to the wavefront bodies' algebra — buffer reuse, in-place inverse, transposed products, every flag — before a GPU is involved.  What this cannot
see is a race between lanes; tests/test_tree_engine_gpu.py runs the same graphs at d > 8 on the device.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rxinfer.jl_amd", "csrc")
EMUL = os.path.join(ROOT, "tests", "host_emul")

(OP_DERIVE_MUL, OP_DERIVE_ADD, OP_LEAF, OP_NOISE, OP_MUL_OUT, OP_MUL_IN, OP_ADD_OUT, OP_ADD_IN, OP_SHIFT, OP_PRODUCT, OP_MARGINAL, OP_FE_NOISE2, OP_FE_NOISE1,
 OP_FE_NOISE0, OP_FE_ENT, OP_FE_ADD2, OP_SUM_TERMS, OP_PREC_UPDATE) = range(1, 19)
OP_FE_NOISE2M, OP_MARG_PUSH, OP_FE_NOISE_MF = 19, 20, 21
W_OP, W_D0, W_D1, W_OUT, W_IN0, W_IN1, W_IN2, W_FLAGS, W_C0, W_C1, W_VAL, W_VAL2, W_PREC, W_TERM, W_N, W_LIST = range(16)
F_IN0_WP, F_IN1_WP, F_IN2_WP, F_OUT_WP, F_VAL_SLOT, F_VAL2_SLOT, F_NEG, F_STAT, F_RAND_IS_MU = 1, 2, 4, 8, 16, 32, 64, 128, 256
F_PUSH_A, F_PUSH_B, F_FOLD_ENT, F_VAL_MARG = 1024, 2048, 4096, 8192
F_MAY_MISS = 16384


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("wave_emul") / "wave_diff.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-DRXHIP_HOST_EMUL", "-Wno-unknown-pragmas", "-I", EMUL, "-I", CSRC, "-o", so,
                    os.path.join(EMUL, "wave_diff.cpp")], check=True)
    L = ctypes.CDLL(so)
    L.emul_run.restype = ctypes.c_int
    return L


class State:
    """Slot allocator and storage in the executor's layout: element k of slot `off` of replica r at (off + k) * RS + r."""

    def __init__(self, R, seed):
        self.R, self.RS = R, (R + 15) // 16 * 16
        self.rng = np.random.default_rng(seed)
        self.size = dict(msg=0, marg=0, val=0, prec=0, term=0, stat=0)
        self.fill = dict(msg=[], marg=[], val=[], prec=[], term=[], stat=[])
        self.cpool, self.aux, self.ops = [], [], []

    def slot(self, kind, n, values=None):   # values: [R, n]
        off = self.size[kind]
        self.size[kind] += n
        if values is not None:
            self.fill[kind].append((off, np.asarray(values, float).reshape(self.R, n)))
        return off

    def const(self, values):
        off = len(self.cpool)
        self.cpool.extend(np.asarray(values, float).ravel().tolist())
        return off

    def spd(self, d, scale=1.0):
        A = self.rng.standard_normal((self.R, d, d + 2))
        return scale * (A @ A.transpose(0, 2, 1) / (d + 2) + 0.3 * np.eye(d))

    def message(self, d):   # a slot with a random (vector, SPD matrix) pair: valid in either form
        v = self.rng.standard_normal((self.R, d))
        M = self.spd(d)
        il = np.tril_indices(d)
        return self.slot("msg", d + d * (d + 1) // 2, np.concatenate([v, M[:, il[0], il[1]]], axis=1))

    def marginal(self, d):
        v = self.rng.standard_normal((self.R, d))
        M = self.spd(d)
        il = np.tril_indices(d)
        ld = np.linalg.slogdet(M)[1][:, None]
        return self.slot("marg", d + d * (d + 1) // 2 + 1, np.concatenate([v, M[:, il[0], il[1]], ld], axis=1))

    def noise_const(self, d):
        S = self.spd(d)[0]
        W = np.linalg.inv(S)
        return self.const(np.concatenate([S.ravel(), W.ravel(), [np.linalg.slogdet(W)[1]]]))

    def precision_state(self, d):   # [nu | V tri | What d*d | What^-1 d*d | E log|W|]
        V = self.spd(d, 0.1)
        nu = d + 2.0 + self.rng.random((self.R, 1))
        What = nu[:, :, None] * V
        il = np.tril_indices(d)
        el = self.rng.standard_normal((self.R, 1))
        return self.slot("prec", 2 + d * (d + 1) // 2 + 2 * d * d, np.concatenate([nu, V[:, il[0], il[1]], What.reshape(self.R, -1), np.linalg.inv(What).reshape(self.R, -1), el], axis=1))

    def op(self, code, d, **kw):
        w = [0] * 16
        w[W_OP], w[W_D0] = code, d
        for k in (W_IN0, W_IN1, W_IN2, W_OUT, W_PREC, W_TERM):
            w[k] = -1
        for k, v in kw.items():
            w[globals()["W_" + k.upper()]] = v
        self.ops.append(w)

    def arrays(self):
        out = {}
        for kind, n in self.size.items():
            a = np.zeros((max(n, 1), self.RS))
            for off, vals in self.fill[kind]:
                a[off:off + vals.shape[1], :self.R] = vals.T
            out[kind] = a
        return out


def run(lib, st, which, n, want_fe=1):
    arr = st.arrays()
    ops = np.asarray(st.ops, np.int32).ravel()
    aux = np.asarray(st.aux + [0], np.int32)
    cp = np.asarray(st.cpool + [0.0], float)
    ptr = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    status = lib.emul_run(which, n, ptr(ops, ctypes.c_int), len(st.ops), ptr(aux, ctypes.c_int), ptr(cp, ctypes.c_double),
                          *[ptr(arr[k], ctypes.c_double) for k in ("msg", "marg", "val", "prec", "term", "stat")],
                          ctypes.c_longlong(st.R), ctypes.c_longlong(st.RS), want_fe)
    return status, arr


def build_program(d, d1, seed):
    """Every opcode with every flag the compiler emits, on independent inputs; (d, d1) the dimensions of a `*` node's out and in."""
    st = State(R=3, seed=seed)
    msz = lambda n: n + n * (n + 1) // 2
    new_msg = lambda n: st.slot("msg", msz(n))
    rng = st.rng
    # clamped values: data slots and constants, and the deterministic nodes over them
    x1 = st.slot("val", d1, rng.standard_normal((st.R, d1)))
    y0 = st.slot("val", d, rng.standard_normal((st.R, d)))
    cv = st.const(rng.standard_normal(d))
    A = st.const(rng.standard_normal((d, d1)))
    dv = st.slot("val", d)
    st.op(OP_DERIVE_MUL, d, d1=d1, out=dv, c0=A, val=x1, flags=F_VAL_SLOT)
    dv2 = st.slot("val", d)
    st.op(OP_DERIVE_ADD, d, out=dv2, val=dv, val2=cv, flags=F_VAL_SLOT)
    noise, ps = st.noise_const(d), st.precision_state(d)
    # leaves
    for wp in (0, F_OUT_WP):
        st.op(OP_LEAF, d, out=new_msg(d), val=y0, flags=F_VAL_SLOT | wp, c0=noise)
        st.op(OP_LEAF, d, out=new_msg(d), val=cv, flags=wp, prec=ps)
    # the additive rule in both forms, constant noise and a precision variable's
    for wp in (0, F_IN0_WP | F_OUT_WP, F_OUT_WP):   # (moments in, precision form out: the conversion done once at the producer)
        st.op(OP_NOISE, d, in0=st.message(d), out=new_msg(d), flags=wp, c0=noise)
        st.op(OP_NOISE, d, in0=st.message(d), out=new_msg(d), flags=wp, prec=ps)
    for wp in (0, F_IN0_WP):
        st.op(OP_MUL_OUT, d, d1=d1, in0=st.message(d1), out=new_msg(d), c0=A, flags=wp)
        st.op(OP_MUL_IN, d, d1=d1, in0=st.message(d), out=new_msg(d1), c0=A, flags=wp)
    for f in (0, F_IN0_WP, F_IN1_WP, F_IN0_WP | F_IN1_WP):
        st.op(OP_ADD_OUT, d, in0=st.message(d), in1=st.message(d), out=new_msg(d), flags=f)
        st.op(OP_ADD_IN, d, in0=st.message(d), in1=st.message(d), out=new_msg(d), flags=f)
    for f in (0, F_IN0_WP, F_NEG, F_IN0_WP | F_NEG):
        st.op(OP_SHIFT, d, in0=st.message(d), out=new_msg(d), val=y0, flags=f | F_VAL_SLOT)
        st.op(OP_SHIFT, d, in0=st.message(d), out=new_msg(d), val=cv, flags=f)
    # products and marginals of three messages in mixed forms
    for code in (OP_PRODUCT, OP_MARGINAL):
        lst = len(st.aux)
        for form in (1, 0, 1):
            st.aux += [st.message(d), form]
        st.op(code, d, out=new_msg(d) if code == OP_PRODUCT else st.slot("marg", msz(d) + 1), n=3, list=lst)
    # the marginal of ONE message: in moment form it is the message itself (no round trip through the precision), in precision form one inverse
    st.single = []
    for form in (0, 1):
        lst = len(st.aux)
        st.aux += [st.message(d), form]
        st.op(OP_MARGINAL, d, out=st.slot("marg", msz(d) + 1), n=1, list=lst)
        st.single.append((st.aux[lst], st.ops[-1][W_OUT], form))
    # … and next to `missing` observations (zeros of the precision form) in a graph created with allow_missing: still the one moment-form message
    zero = st.message(d)
    st.fill["msg"][-1][1][:] = 0.0
    lst = len(st.aux)
    st.aux += [zero, 1, st.message(d), 0, zero, 1]
    st.op(OP_MARGINAL, d, out=st.slot("marg", msz(d) + 1), n=3, list=lst, flags=F_MAY_MISS)
    st.single.append((st.aux[lst + 2], st.ops[-1][W_OUT], 0))
    lst = len(st.aux)
    st.aux += [zero, 1, st.message(d), 0, st.message(d), 1]   # (not the case: a precision-form message with content)
    st.op(OP_MARGINAL, d, out=st.slot("marg", msz(d) + 1), n=3, list=lst, flags=F_MAY_MISS)
    # second phase
    terms = []
    new_term = lambda: terms.append(st.slot("term", 1)) or terms[-1]
    stats = []
    for f in (0, F_IN0_WP, F_IN1_WP | F_IN0_WP):
        st.op(OP_FE_NOISE2, d, in0=st.message(d), in1=st.message(d), flags=f, c0=noise, term=new_term())
        stats.append(st.slot("stat", d * d))
        st.op(OP_FE_NOISE2, d, in0=st.message(d), in1=st.message(d), flags=f | F_STAT, prec=ps, c1=stats[-1], term=new_term())
    st.op(OP_FE_NOISE2, d, in0=-1, in1=st.message(d), flags=F_IN1_WP, c0=noise, term=new_term())
    st.op(OP_FE_NOISE2, d, in0=st.message(d), in1=-1, flags=0, prec=ps, term=new_term())
    mg = st.marginal(d)
    st.op(OP_FE_NOISE1, d, in0=mg, val=y0, flags=F_VAL_SLOT, c0=noise, term=new_term())
    stats.append(st.slot("stat", d * d))
    st.op(OP_FE_NOISE1, d, in0=mg, val=cv, flags=F_STAT | F_RAND_IS_MU, prec=ps, c1=stats[-1], term=new_term())
    st.op(OP_FE_NOISE0, d, val=y0, val2=cv, flags=F_VAL_SLOT, c0=noise, term=new_term())
    stats.append(st.slot("stat", d * d))
    st.op(OP_FE_NOISE0, d, val=cv, val2=y0, flags=F_VAL2_SLOT | F_STAT, prec=ps, c1=stats[-1], term=new_term())
    st.op(OP_FE_ENT, d, in0=mg, n=-2, term=new_term())
    # round 6: the one-message joint term with stored and with image marginals (the marginal of `A * x` as the image of x's), entropy folded in; the stored
    # marginal of an image; the image in the one-interface term and in the entropy term; a Gaussian node under q(out) q(μ) — its leaf rule and average energy
    mg2, mgs = st.marginal(d), st.marginal(d)   # (images through a SQUARE map: with more rows than columns the image is singular and its log-determinant −∞ by design)
    A2m = rng.standard_normal((d, d))
    A2 = st.const(A2m)
    pairs = []   # (term computed with a Cholesky of the image, the same term with log|A V A'| = log|V| + 2 log|det A| from the constant pool)
    for ld in (-1, st.const(np.array([2.0 * np.linalg.slogdet(A2m)[1]]))):
        these = []
        term = lambda: these.append(new_term()) or these[-1]
        msgs = [st.message(d) for _ in range(2)] if ld < 0 else msgs
        for f, m in zip((0, F_IN0_WP), msgs):
            st.op(OP_FE_NOISE2M, d, d1=ld, in0=m, val=mg2, val2=mgs, in2=A2, n=d, flags=f | F_PUSH_B | F_FOLD_ENT, out=1, c0=noise, term=term())
            st.op(OP_FE_NOISE2M, d, d1=-1, in0=m, val=mgs, in1=A2, list=d, val2=mg, flags=f | F_PUSH_A, c0=noise, term=new_term(), out=0)
            stats.append(st.slot("stat", d * d))
            st.op(OP_FE_NOISE2M, d, d1=ld, in0=m, val=mg, val2=mgs, in2=A2, n=d, flags=f | F_PUSH_B | F_STAT, prec=ps, c1=stats[-1], term=term(), out=0)
        st.op(OP_MARG_PUSH, d, d1=d, in0=mgs, in1=ld, c0=A2, out=st.slot("marg", msz(d) + 1))
        these.append(("marg", st.ops[-1][W_OUT] + msz(d)))
        st.op(OP_FE_NOISE1, d, d1=d, in0=mgs, in1=A2, in2=ld, val=y0, flags=F_VAL_SLOT | F_PUSH_A | F_FOLD_ENT, out=2, c0=noise, term=term())
        st.op(OP_FE_ENT, d, d1=d, in0=mgs, in1=ld, c0=A2, n=3, flags=F_PUSH_A, term=term())
        pairs.append(these)
    st.same = list(zip(*pairs))
    for f in (0, F_IN0_WP):
        st.op(OP_FE_NOISE2M, d, in0=st.message(d), val=mg, val2=mg2, flags=f, c0=noise, term=new_term(), out=0)
    st.op(OP_FE_NOISE2M, d, in0=-1, val=mg, val2=mg2, flags=0, c0=noise, term=new_term(), out=0)
    st.op(OP_FE_NOISE_MF, d, val=mg, val2=mg2, c0=noise, term=new_term())
    stats.append(st.slot("stat", d * d))
    st.op(OP_FE_NOISE_MF, d, val=mg2, val2=mg, flags=F_STAT, prec=ps, c1=stats[-1], term=new_term())
    for wp in (0, F_OUT_WP):
        st.op(OP_LEAF, d, out=new_msg(d), val=mg, flags=F_VAL_MARG | wp, c0=noise)
    for f, ins in ((0, (1, 1, 1)), (F_IN0_WP | F_IN2_WP, (1, 1, 1)), (F_IN1_WP, (0, 1, 1)), (F_IN0_WP, (1, 0, 1)), (F_IN0_WP | F_IN1_WP, (1, 1, 0))):
        slots = [st.message(d) if k else -1 for k in ins]
        st.op(OP_FE_ADD2, d, in0=slots[0], in1=slots[1], in2=slots[2], flags=f, term=new_term())
    # q(W) update from the residual moments written above, into a second precision state
    S0 = st.spd(d, 0.2)[0]
    prior = st.const(np.concatenate([[d + 1.5], np.linalg.inv(S0).ravel(), [np.linalg.slogdet(S0)[1]]]))
    ps2 = st.slot("prec", 2 + d * (d + 1) // 2 + 2 * d * d)
    lst = len(st.aux)
    st.aux += stats
    st.op(OP_PREC_UPDATE, d, c0=prior, prec=ps2, n=len(stats), list=lst, term=new_term())
    lst = len(st.aux)
    st.aux += terms
    st.op(OP_SUM_TERMS, d, n=len(terms), list=lst, term=st.slot("term", 1))
    return st


@pytest.mark.parametrize("d,d1", [(3, 2), (2, 3), (8, 5), (12, 9), (9, 16), (20, 20), (32, 24), (33, 40), (64, 48)])
def test_every_op_matches_the_register_bodies(lib, d, d1):
    st = build_program(d, d1, seed=100 * d + d1)
    n = max(d, d1)
    s0, ref = run(lib, st, 0, n)
    s1, got = run(lib, st, 1, n)
    assert s0 == 0 and s1 == 0
    for kind in ("msg", "marg", "val", "prec", "term", "stat"):
        a, b = ref[kind][:, :st.R], got[kind][:, :st.R]
        assert np.all(np.isfinite(a)) and np.all(np.isfinite(b)), kind
        scale = np.maximum(1.0, np.abs(a))
        err = np.max(np.abs(a - b) / scale)
        assert err < 1e-11 * n, (kind, err)
        assert np.any(a != 0.0) or st.size[kind] == 0, kind
    # the marginal of one moment-form message is that message, bit for bit; its last word the log-determinant of the covariance
    msz = lambda k: k + k * (k + 1) // 2
    for which in (ref, got):
        for src, out, form in st.single:
            m, g = which["msg"][src:src + msz(d), :st.R], which["marg"][out:out + msz(d) + 1, :st.R]
            if form == 0:
                assert np.array_equal(m, g[:-1])
            for r in range(st.R):
                M = np.zeros((d, d))
                M[np.tril_indices(d)] = (g if form == 0 else m)[d:msz(d), r]
                M = M + M.T - np.diag(np.diag(M))
                ld = np.linalg.slogdet(M)[1]
                assert abs(g[-1, r] - (ld if form == 0 else -ld)) < 1e-9 * n
    # a square map's image: the log-determinant by addition gives the term the Cholesky of the image gives
    for which in (ref, got):
        for a, b in st.same:
            ka, oa = a if isinstance(a, tuple) else ("term", a)
            kb, ob = b if isinstance(b, tuple) else ("term", b)
            va, vb = which[ka][oa, :st.R], which[kb][ob, :st.R]
            assert np.allclose(va, vb, rtol=1e-10, atol=1e-10 * n), (a, b, va, vb)


def test_a_matrix_that_is_not_positive_definite_is_reported(lib):
    st = State(R=2, seed=5)
    d = 12
    off = st.message(d)
    st.fill["msg"][-1][1][:, d] = -1.0   # first diagonal entry of the matrix
    st.op(OP_MARGINAL, d, out=st.slot("marg", d + d * (d + 1) // 2 + 1), n=1, list=0)
    st.aux += [off, 1]
    assert run(lib, st, 0, d)[0] == 1
    assert run(lib, st, 1, d)[0] == 1


@pytest.mark.parametrize("d,which", [(3, 0), (3, 1), (12, 0), (12, 1)])
def test_backward_rule_of_plus_in_precision_form(lib, d, which):
    """typeof(+)(:in) on a precision-form message from `out`: the same Gaussian as the moment-form rule N(m_out − m2, V_out + V2), and finite where the
    message from `out` is rank-deficient (where it equals the limit of the proper case)"""
    st = State(R=2, seed=9 + d)
    rng = st.rng
    il = np.tril_indices(d)
    pack = lambda v, M: np.concatenate([v, M[:, il[0], il[1]]], axis=1)
    mo, Vo, m2, V2 = rng.standard_normal((2, d)), st.spd(d), rng.standard_normal((2, d)), st.spd(d)
    Lo, W2 = np.linalg.inv(Vo), np.linalg.inv(V2)
    msz = d + d * (d + 1) // 2
    a_mv, b_mv = st.slot("msg", msz, pack(mo, Vo)), st.slot("msg", msz, pack(m2, V2))
    a_wp, b_wp = st.slot("msg", msz, pack(np.einsum("rij,rj->ri", Lo, mo), Lo)), st.slot("msg", msz, pack(np.einsum("rij,rj->ri", W2, m2), W2))
    # a rank-deficient message from `out`: B' Q⁻¹ B of a map with one row
    B = rng.standard_normal((1, d))
    Ld = np.broadcast_to(B.T @ B, (2, d, d)).copy()
    xd = rng.standard_normal((2, 1)) * B
    a_def = st.slot("msg", msz, pack(xd, Ld))
    outs = [st.slot("msg", msz) for _ in range(4)]
    st.op(OP_ADD_IN, d, in0=a_mv, in1=b_mv, out=outs[0], flags=0)
    st.op(OP_ADD_IN, d, in0=a_wp, in1=b_wp, out=outs[1], flags=F_IN0_WP | F_IN1_WP)
    st.op(OP_ADD_IN, d, in0=a_wp, in1=b_mv, out=outs[2], flags=F_IN0_WP)
    st.op(OP_ADD_IN, d, in0=a_def, in1=b_mv, out=outs[3], flags=F_IN0_WP)
    status, arr = run(lib, st, which, d)
    assert status == 0

    def unpack(off):
        blk = arr["msg"][off:off + msz, :2].T
        M = np.zeros((2, d, d))
        M[:, il[0], il[1]] = blk[:, d:]
        M = M + np.transpose(np.tril(M, -1), (0, 2, 1))
        return blk[:, :d], M
    m_ref, V_ref = unpack(outs[0])
    assert np.allclose(m_ref, mo - m2, rtol=1e-13) and np.allclose(V_ref, Vo + V2, rtol=1e-13)
    for o in outs[1:3]:
        xi, L = unpack(o)
        V = np.linalg.inv(L)
        assert np.allclose(V, V_ref, rtol=1e-9, atol=1e-11) and np.allclose(np.einsum("rij,rj->ri", V, xi), m_ref, rtol=1e-9, atol=1e-10)
    # rank-deficient: against the closed form evaluated in numpy, and rank 1 as it must be (Λ' = Λo (Λo + W2)⁻¹ W2)
    xi, L = unpack(outs[3])
    G = np.linalg.inv(Ld + W2)
    x2 = np.einsum("rij,rj->ri", W2, m2)
    assert np.allclose(L, Ld @ G @ W2, rtol=1e-10, atol=1e-12) and np.allclose(xi, np.einsum("rij,rj->ri", W2 @ G, xd + x2) - x2, rtol=1e-10, atol=1e-11)
    assert np.all(np.linalg.matrix_rank(L, tol=1e-9) == 1)

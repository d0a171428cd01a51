"""The algebra of csrc/dense_mseg_kernels.hpp, restated in numpy and checked on the CPU against the oracle's smoother.

A segment (x_b -> x_e, with its transitions and its observed / missing observations) is carried as the joint information of (x_b, x_e):
precision [[Ĵ, −Ψ′], [−Ψ, Λ]], vector [η̂, ξ].  One more time step = add the transition factor, eliminate the old end state, add the new
observation's information (km_elements); the filtered belief travels across a segment by adding it to the x_b corner and eliminating x_b,
the backward message by adding it to the x_e corner and eliminating x_e (km_scan).  The test builds the elements of a ragged segmentation
of a chain with missing observations, runs both boundary recursions, and compares
  * the filtered belief at every boundary with a plain Kalman filter,
  * (Λ_f + Λβ)⁻¹ and its mean at every boundary with the oracle's smoothed posterior."""
import numpy as np
import pytest


def _model(d, dy, seed):
    rng = np.random.default_rng(seed)
    Qm, _ = np.linalg.qr(rng.standard_normal((d, d)))
    A = 0.95 * Qm
    B = rng.standard_normal((dy, d)) / np.sqrt(d)
    Wp = rng.standard_normal((d, d)) * 0.3
    P = Wp @ Wp.T + 0.1 * np.eye(d)
    Wq = rng.standard_normal((dy, dy)) * 0.5
    Q = Wq @ Wq.T + 0.5 * np.eye(dy)
    Wv = rng.standard_normal((d, d))
    return A, B, P, Q, rng.standard_normal(d), Wv @ Wv.T + np.eye(d)


def _element(A, B, P, Q, ys):
    """joint information of (x_b, x_e) for the steps whose observations are `ys` (NaN row = missing): km_elements"""
    Pi, Qi = np.linalg.inv(P), np.linalg.inv(Q)
    K, W, Lobs, G = Pi @ A, A.T @ Pi @ A, B.T @ Qi @ B, B.T @ Qi
    lam = psi = jh = xi = eta = None
    for y in ys:
        ob = not np.any(np.isnan(y))
        if lam is None:                       # out of the known start through the first transition
            lam_p, psi, jh, xi_p, eta = Pi.copy(), K.copy(), W.copy(), np.zeros(len(A)), np.zeros(len(A))
        else:
            C = np.linalg.inv(lam + W)
            Y, c = C @ psi, C @ xi
            lam_p = Pi - K @ C @ K.T
            jh = jh - psi.T @ Y
            eta = eta + psi.T @ c
            psi = K @ Y
            xi_p = K @ c
        lam = lam_p + (Lobs if ob else 0.0)
        xi = xi_p + (G @ y if ob else 0.0)
    return lam, psi, jh, xi, eta


@pytest.mark.parametrize("d,dy,T,L,seed", [(3, 2, 41, 7, 1), (6, 6, 30, 4, 2), (5, 1, 26, 25, 3), (4, 3, 12, 1, 4)])
def test_information_form_elements_and_boundary_recursions(d, dy, T, L, seed):
    import rxoracle as rxo
    A, B, P, Q, m0, V0 = _model(d, dy, seed)
    rng = np.random.default_rng(seed + 10)
    x = rng.multivariate_normal(m0, V0)
    y = np.empty((T, dy))
    for t in range(T):
        if t:
            x = A @ x + rng.multivariate_normal(np.zeros(d), P)
        y[t] = B @ x + rng.multivariate_normal(np.zeros(dy), Q)
    y[rng.random(T) < 0.25] = np.nan
    y[0] = np.nan                                        # first observation missing
    if T > 20:
        y[8:16] = np.nan                                 # a whole segment (and more) without observations
    om, oc, _ = rxo.lgssm_kalman_rts(A, B, P, Q, m0, V0, np.ascontiguousarray(y))
    # plain Kalman filter (covariance form) for the filtered beliefs
    Qi = np.linalg.inv(Q)
    mf, Vf = [], []
    m, V = m0, V0
    for t in range(T):
        if t:
            m, V = A @ m, A @ V @ A.T + P
        if not np.any(np.isnan(y[t])):
            S = B @ V @ B.T + Q
            Kg = V @ B.T @ np.linalg.inv(S)
            m, V = m + Kg @ (y[t] - B @ m), V - Kg @ B @ V
        mf.append(m); Vf.append(V)
    # segments: boundaries b_s = s·L, segment s covers the steps b_s + 1 … min(b_s + L, T − 1)
    bnd = list(range(0, T - 1, L)) + [T - 1]
    els = [_element(A, B, P, Q, y[bnd[s] + 1: bnd[s + 1] + 1]) for s in range(len(bnd) - 1)]
    # prefix: filtered belief at every boundary
    ob0 = not np.any(np.isnan(y[0]))
    lam_f = np.linalg.inv(V0) + (B.T @ Qi @ B if ob0 else 0.0)
    xi_f = np.linalg.inv(V0) @ m0 + (B.T @ Qi @ y[0] if ob0 else 0.0)
    pre = [(lam_f, xi_f)]
    for lam, psi, jh, xi, eta in els:
        Ti = np.linalg.inv(lam_f + jh)
        lam_f, xi_f = lam - psi @ Ti @ psi.T, xi + psi @ Ti @ (xi_f + eta)
        pre.append((lam_f, xi_f))
    for s, (lf, xf) in enumerate(pre):
        V = np.linalg.inv(lf)
        assert np.allclose(V, Vf[bnd[s]], rtol=1e-9, atol=1e-11), s
        assert np.allclose(V @ xf, mf[bnd[s]], rtol=1e-9, atol=1e-11), s
    # suffix: backward message at every boundary, then the smoothed belief there
    lam_b, xi_b = np.zeros((d, d)), np.zeros(d)
    suf = [(lam_b, xi_b)]
    for lam, psi, jh, xi, eta in reversed(els):
        Ti = np.linalg.inv(lam + lam_b)
        lam_b, xi_b = jh - psi.T @ Ti @ psi, eta + psi.T @ Ti @ (xi + xi_b)
        suf.append((lam_b, xi_b))
    suf.reverse()
    for s in range(len(bnd)):
        Vs = np.linalg.inv(pre[s][0] + suf[s][0])
        assert np.allclose(Vs, oc[bnd[s]], rtol=1e-8, atol=1e-10), s
        assert np.allclose(Vs @ (pre[s][1] + suf[s][1]), om[bnd[s]], rtol=1e-8, atol=1e-10), s


def _compose(e1, e2):
    """two segments in a row as one: stack the joints of (x_a, x_b) and (x_b, x_c), eliminate x_b (km_group)"""
    l1, p1, j1, x1, h1 = e1
    l2, p2, j2, x2, h2 = e2
    Ti = np.linalg.inv(l1 + j2)
    u = x1 + h2
    return l2 - p2 @ Ti @ p2.T, p2 @ Ti @ p1, j1 - p1.T @ Ti @ p1, x2 + p2 @ Ti @ u, h1 + p1.T @ Ti @ u


@pytest.mark.parametrize("d,dy,n1,n2,seed", [(4, 2, 5, 3, 1), (7, 7, 1, 9, 2), (3, 1, 6, 1, 3)])
def test_composition_of_two_elements_is_the_element_of_both(d, dy, n1, n2, seed):
    A, B, P, Q, _, _ = _model(d, dy, seed)
    rng = np.random.default_rng(seed)
    ys = rng.standard_normal((n1 + n2, dy))
    ys[rng.random(n1 + n2) < 0.3] = np.nan
    whole = _element(A, B, P, Q, ys)
    both = _compose(_element(A, B, P, Q, ys[:n1]), _element(A, B, P, Q, ys[n1:]))
    for a, b in zip(whole, both):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-11)


def _generations(i):
    """rounds an entry at distance i from the origin of its scan takes part in (hs_generations)"""
    return 0 if i == 0 else int(i).bit_length()


@pytest.mark.parametrize("S", [2, 3, 4, 5, 8, 9, 17, 31, 32, 33])
def test_log_depth_rounds_with_two_buffers_give_every_prefix_and_suffix(S):
    """km_compose / km_apply: ⌈log₂⌉ rounds of pairwise compositions, generation g ≥ 1 of an entry kept in buffer (g − 1) mod 2 (generation 0:
    the element itself), the whole chain's composition never formed — every P[j] = E_0 ∘ … ∘ E_j (j ≤ S − 2) and Q[j] = E_j ∘ … ∘ E_{S−1}
    (j ≥ 1) must come out where km_apply looks for it, and nothing may be read from a slot that the same round writes."""
    d, dy = 3, 2
    A, B, P, Q, _, _ = _model(d, dy, S)
    rng = np.random.default_rng(S)
    els = []
    for s in range(S):
        ys = rng.standard_normal((int(rng.integers(1, 4)), dy))
        ys[rng.random(len(ys)) < 0.3] = np.nan
        els.append(_element(A, B, P, Q, ys))
    rounds = 0
    while (1 << rounds) <= S - 2:
        rounds += 1
    buf = {}                                     # (dir, parity, index) -> element

    def get(dr, idx, gen):
        return els[idx] if gen == 0 else buf[(dr, (gen - 1) & 1, idx)]
    for r in range(rounds):
        h = 1 << r
        written, read = {}, set()
        for dr in (0, 1):
            for j in range(S):
                i = j if dr == 0 else S - 1 - j
                if i < h or i == S - 1:
                    continue
                jp = j - h if dr == 0 else j + h
                gp = min(_generations(i - h), r)
                e_self, e_part = get(dr, j, r), get(dr, jp, gp)
                if r > 0:
                    read.add((dr, (r - 1) & 1, j))
                if gp > 0:
                    read.add((dr, (gp - 1) & 1, jp))
                written[(dr, r & 1, j)] = _compose(e_part, e_self) if dr == 0 else _compose(e_self, e_part)
        assert not (read & set(written)), (S, r)
        buf.update(written)
    for j in range(S - 1):                       # prefixes that km_apply reads
        ref = els[0]
        for e in els[1:j + 1]:
            ref = _compose(ref, e)
        for a, b in zip(get(0, j, _generations(j)), ref):
            assert np.allclose(a, b, rtol=1e-8, atol=1e-10), (S, j)
    for j in range(1, S):                        # suffixes
        ref = els[S - 1]
        for e in reversed(els[j:S - 1]):
            ref = _compose(e, ref)
        for a, b in zip(get(1, j, _generations(S - 1 - j)), ref):
            assert np.allclose(a, b, rtol=1e-8, atol=1e-10), (S, j)


@pytest.mark.parametrize("S,g", [(7, 2), (9, 4), (12, 3), (5, 8), (16, 4)])
def test_groups_of_segments_fold_scan_and_inner_steps(S, g):
    """km_fold / km_inner around the rounds: fold every group of g segments into one entry, take the boundary states at the group edges from
    the prefix / suffix compositions of the entries, carry them inwards with one boundary step per segment — the same states as the plain
    recursions over all segments (ragged last group included)"""
    d, dy = 3, 2
    A, B, P, Q, m0, V0 = _model(d, dy, 10 * S + g)
    rng = np.random.default_rng(S + g)
    els = []
    for s in range(S):
        ys = rng.standard_normal((int(rng.integers(1, 4)), dy))
        ys[rng.random(len(ys)) < 0.3] = np.nan
        els.append(_element(A, B, P, Q, ys))

    def fwd(state, e):
        lam, psi, jh, xi, eta = e
        Ti = np.linalg.inv(state[0] + jh)
        return lam - psi @ Ti @ psi.T, xi + psi @ Ti @ (state[1] + eta)

    def bwd(e, state):
        lam, psi, jh, xi, eta = e
        Ti = np.linalg.inv(lam + state[0])
        return jh - psi.T @ Ti @ psi, eta + psi.T @ Ti @ (xi + state[1])
    b0 = (np.linalg.inv(V0), np.linalg.inv(V0) @ m0)
    zero = (np.zeros((d, d)), np.zeros(d))
    pre, suf = [b0], [zero]
    for e in els:
        pre.append(fwd(pre[-1], e))
    for e in reversed(els):
        suf.append(bwd(e, suf[-1]))
    suf.reverse()                                  # suf[s]: message at the START of segment s = at the end of segment s − 1
    n = (S + g - 1) // g
    ent = []
    for k in range(n):                             # km_fold
        run = els[g * k]
        for e in els[g * k + 1: min(g * k + g, S)]:
            run = _compose(run, e)
        ent.append(run)
    for k in range(n):
        s0, s1 = g * k, min(g * k + g, S)
        # km_apply: belief at the start of the group through the prefix composition of the entries in front of it …
        state = b0
        if k:
            P_ = ent[0]
            for e in ent[1:k]:
                P_ = _compose(P_, e)
            state = fwd(b0, P_)
        # … and km_inner from there
        for s in range(s0, s1):
            assert np.allclose(state[0], pre[s][0], rtol=1e-8, atol=1e-10) and np.allclose(state[1], pre[s][1], rtol=1e-8, atol=1e-10), (k, s)
            state = fwd(state, els[s])
        state = zero
        if k < n - 1:
            Q_ = ent[n - 1]
            for e in reversed(ent[k + 1: n - 1]):
                Q_ = _compose(e, Q_)
            state = bwd(Q_, zero)
        for s in range(s1 - 1, s0 - 1, -1):        # message at the END of segment s
            assert np.allclose(state[0], suf[s + 1][0], rtol=1e-8, atol=1e-10) and np.allclose(state[1], suf[s + 1][1], rtol=1e-8, atol=1e-10), (k, s)
            state = bwd(els[s], state)


def _element_fused(A, B, P, Q, ys):
    """the step as the fused kernel forms it (km_elements): with C in the staging matrix, K C and Y′ = Ψ′C in one pass, the vectors as row
    sums of those two products against the old ξ, Ĵ −= Ψ′Y and Ψ = K Y from the staged Y′, Λp = P⁻¹ − K (K C)′ symmetrised"""
    Pi, Qi = np.linalg.inv(P), np.linalg.inv(Q)
    Pi = 0.5 * (Pi + Pi.T)
    K, W, Lobs, G = Pi @ A, A.T @ Pi @ A, B.T @ Qi @ B, B.T @ Qi
    W, Lobs = 0.5 * (W + W.T), 0.5 * (Lobs + Lobs.T)          # the symmetric copies kt_consts leaves
    lam = None
    for y in ys:
        ob = not np.any(np.isnan(y))
        gy = G @ y if ob else np.zeros(len(A))
        if lam is None:
            lam, psi, jh, xi, eta = Pi.copy(), K.copy(), W.copy(), gy, np.zeros(len(A))
        else:
            C = np.linalg.inv(lam + W + (Lobs if ob_prev else 0.0))
            gt, yt = K @ C, psi.T @ C
            xi, eta = gt @ xi + gy, eta + yt @ xi
            jh, psi = jh - psi.T @ yt.T, K @ yt.T
            lp = Pi - K @ gt.T
            lam = 0.5 * (lp + lp.T)
        ob_prev = ob
    return lam + (Lobs if ob_prev else 0.0), psi, jh, xi, eta


def _compose_fused(e1, e2):
    """mseg_compose_fused: A′ = Ψ1′T⁻¹ and B = Ψ2 T⁻¹ from the staged T⁻¹, the vectors as their row sums, then Ψ1′A, Ψ2 A, Ψ2 B′"""
    l1, p1, j1, x1, h1 = e1
    l2, p2, j2, x2, h2 = e2
    Ti = np.linalg.inv(l1 + j2)
    at, bq, u = p1.T @ Ti, p2 @ Ti, x1 + h2
    t2 = p2 @ bq.T
    return l2 - 0.5 * (t2 + t2.T), p2 @ at.T, j1 - p1.T @ at.T, x2 + bq @ u, h1 + at @ u


@pytest.mark.parametrize("d,dy,n,seed", [(4, 2, 7, 1), (6, 6, 5, 2), (3, 1, 9, 3)])
def test_fused_forms_are_the_same_algebra(d, dy, n, seed):
    A, B, P, Q, _, _ = _model(d, dy, seed)
    rng = np.random.default_rng(seed)
    ys = rng.standard_normal((2 * n, dy))
    ys[rng.random(2 * n) < 0.3] = np.nan
    for a, b in zip(_element(A, B, P, Q, ys), _element_fused(A, B, P, Q, ys)):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-11)
    e1, e2 = _element(A, B, P, Q, ys[:n]), _element(A, B, P, Q, ys[n:])
    for a, b in zip(_compose(e1, e2), _compose_fused(e1, e2)):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-11)

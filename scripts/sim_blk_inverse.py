"""Lane-level model of the blocked SPD inverse of csrc/dense_kernels.hpp (blk_inverse): every register, LDS slot and
v_mfma_f64_16x16x4_f64 operand is indexed exactly as the kernel indexes it, so the index algebra (accumulator layout reused
as operand layout, rotated Cramer columns, modified-operand fix-ups of the pivot block) is checked on the CPU before a GPU
minute is spent.  Run: python scripts/sim_blk_inverse.py"""
import numpy as np


def mfma(a, b, acc):
    """v_mfma_f64_16x16x4_f64: a[lane], b[lane] doubles, acc[lane][4].  A operand: lane (i = l&15, k = l>>4) holds X[i][k];
    B operand: lane (j = l&15, k = l>>4) holds Y[k][j]; C/D: lane (j = l&15, q = l>>4) reg r holds (row q + 4r, col j)."""
    X = np.zeros((16, 4))
    Y = np.zeros((4, 16))
    for l in range(64):
        X[l & 15, l >> 4] = a[l]
        Y[l >> 4, l & 15] = b[l]
    P = X @ Y
    out = acc.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += P[(l >> 4) + 4 * r, l & 15]
    return out


def diag_inverse16(t, stats):
    """in-wave inverse of one 16×16 SPD tile held in accumulator layout t[lane][4]; returns +D⁻¹ in the same layout, det list, ok"""
    t = t.copy()
    dets = []
    ok = True
    lanes = np.arange(64)
    j, q = lanes & 15, lanes >> 4
    jq, ju = j & 3, j >> 2
    for Q in range(4):
        # owner lanes (q == Q) publish their four pivot-row values: R[u][j] = D[Q + 4u][j]
        sb = np.zeros((16, 4))
        for l in range(64):
            if q[l] == Q:
                sb[j[l], :] = t[l, :]
        xa = np.zeros(64)
        yb = np.zeros(64)
        for l in range(64):
            ri = sb[j[l], :].copy()                      # column i = j of R
            fQ = 1.0 if jq[l] == Q else 0.0
            for u in range(4):
                ri[u] -= fQ * (1.0 if ju[l] == u else 0.0)   # R̃ = R − E_K
            ybl = sb[j[l], q[l]] - fQ * (1.0 if ju[l] == q[l] else 0.0)
            # pivot block, columns rotated so that this lane's own column v = q comes first
            cols = [sb[Q + 4 * ((q[l] + m) & 3), :] for m in range(4)]   # cols[m][u] = D4[u][(q+m)&3]
            a_, b_, c_, d_ = cols
            m01 = c_[0] * d_[1] - c_[1] * d_[0]
            m02 = c_[0] * d_[2] - c_[2] * d_[0]
            m03 = c_[0] * d_[3] - c_[3] * d_[0]
            m12 = c_[1] * d_[2] - c_[2] * d_[1]
            m13 = c_[1] * d_[3] - c_[3] * d_[1]
            m23 = c_[2] * d_[3] - c_[3] * d_[2]
            cof0 = b_[1] * m23 - b_[2] * m13 + b_[3] * m12
            cof1 = -(b_[0] * m23 - b_[2] * m03 + b_[3] * m02)
            cof2 = b_[0] * m13 - b_[1] * m03 + b_[3] * m01
            cof3 = -(b_[0] * m12 - b_[1] * m02 + b_[2] * m01)
            det = a_[0] * cof0 + a_[1] * cof1 + a_[2] * cof2 + a_[3] * cof3
            num = ri[0] * cof0 + ri[1] * cof1 + ri[2] * cof2 + ri[3] * cof3
            xa[l] = -num / det
            yb[l] = ybl
            if q[l] == 0:   # natural column order: nested trailing principal minors
                good = d_[3] > 0 and m23 > 0 and cof0 > 0 and det > 0
                if l == 0:
                    dets.append(det)
                    ok = ok and good
        t = mfma(xa, yb, t)
        stats["mfma_diag"] += 1
    di = np.zeros_like(t)
    for l in range(64):
        for r in range(4):
            di[l, r] = (2.0 if j[l] == q[l] + 4 * r else 0.0) - t[l, r]
    return di, dets, ok


def blk_inverse(A, NT):
    """A: (16 NT)² SPD.  Registers a[w][lane][t][r] ↔ A[16w + q + 4r][16t + j].  Returns the inverse, logdet."""
    D = 16 * NT
    lanes = np.arange(64)
    j, q = lanes & 15, lanes >> 4
    a = np.zeros((NT, 64, NT, 4))
    for w in range(NT):
        for l in range(64):
            for t in range(NT):
                for r in range(4):
                    a[w, l, t, r] = A[16 * w + q[l] + 4 * r, 16 * t + j[l]]
    hexp = np.array([-(int(np.frexp(A[i, i])[1]) >> 1) for i in range(D)])   # s_i = 2^h_i: exact scaling, a_ii s_i² in [1/4, 2)
    for w in range(NT):
        for l in range(64):
            for t in range(NT):
                for r in range(4):
                    a[w, l, t, r] = np.ldexp(a[w, l, t, r], int(hexp[16 * w + q[l] + 4 * r] + hexp[16 * t + j[l]]))
    stats = {"mfma_diag": 0, "mfma_z": 0, "mfma_trail": 0, "mfma_row": 0}
    logdet = 0.0
    for k in range(NT):
        # wave k: invert its diagonal tile, publish the pivot block row R ([t][r][lane]) and −D⁻¹ ([r][lane])
        di, dets, ok = diag_inverse16(a[k, :, k, :], stats)
        assert ok
        logdet += sum(np.log(x) for x in dets)
        Rbuf = a[k].transpose(1, 2, 0).copy()          # [t][r][lane]
        ndi = -di.T.copy()                              # [r][lane]
        # wave k: its own row block, straight from registers: A_kt <- D⁻¹ R_t (t != k), A_kk <- 2I − D⁻¹
        for t in range(NT):
            if t == k:
                for l in range(64):
                    for r in range(4):
                        a[k, l, k, r] = -di[l, r]
            else:
                acc = np.zeros((64, 4))
                src = a[k, :, t, :].copy()
                for r in range(4):
                    acc = mfma(di[:, r], src[:, r], acc)
                    stats["mfma_row"] += 1
                a[k, :, t, :] = acc
        # waves w != k (after the barrier)
        for w in range(NT):
            if w == k:
                continue
            z = np.zeros((64, 4))
            for r in range(4):
                z = mfma(ndi[r], Rbuf[w, r], z)         # Z̃_w = −D⁻¹ R_w, accumulator layout = A-operand layout of Z̃_w'
                stats["mfma_z"] += 1
            for t in range(NT):
                if t == k:   # column block: A_wk <- (D⁻¹R_w)' = R_w'D⁻¹ — the same two operands swapped, no cancellation
                    acc = np.zeros((64, 4))
                    for r in range(4):
                        acc = mfma(Rbuf[w, r], -ndi[r], acc)
                        stats["mfma_trail"] += 1
                else:
                    acc = a[w, :, t, :].copy()
                    for r in range(4):
                        acc = mfma(z[:, r], Rbuf[t, r], acc)
                        stats["mfma_trail"] += 1
                a[w, :, t, :] = acc
    out = np.zeros((D, D))
    for w in range(NT):
        for l in range(64):
            for t in range(NT):
                for r in range(4):
                    v = -a[w, l, t, r]
                    out[16 * w + q[l] + 4 * r, 16 * t + j[l]] = np.ldexp(v, int(hexp[16 * w + q[l] + 4 * r] + hexp[16 * t + j[l]]))
    logdet -= 2.0 * np.log(2.0) * hexp.sum()
    return out, logdet, stats


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for NT in (1, 2, 3, 4):
        D = 16 * NT
        for scale in (1.0, 1e-6, 1e6, None):
            G = rng.standard_normal((D, D))
            A = G @ G.T + D * np.eye(D)
            if scale is None:   # badly scaled coordinates: 12 orders of magnitude between the diagonal entries
                sv = 10.0 ** rng.uniform(-3, 3, D)
                A = A * sv[:, None] * sv[None, :]
            else:
                A = A * scale
            inv, ld, stats = blk_inverse(A, NT)
            ref = np.linalg.inv(A)
            dg = np.sqrt(np.diag(ref))
            err = np.max(np.abs(inv - ref) / (dg[:, None] * dg[None, :]))   # element-wise, in units of sqrt(ref_ii ref_jj)
            lderr = abs(ld - np.linalg.slogdet(A)[1])
            print(f"NT={NT} scale={scale}: err {err:.2e}  logdet err {lderr:.2e}  {stats}")
            assert err < 1e-13 and lderr < 1e-9

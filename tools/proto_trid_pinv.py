"""Prototype (numpy fp64) of the wave-parallel truncated pseudo-inverse solve used by the Gauss-Helmert kernels with redundant
constraints: x = sum_{|lambda_k| > tol} y_k (y_k' b) / lambda_k for a symmetric n x n matrix (n <= 32), by Householder
tridiagonalisation, Sturm counts on geometrically spaced shifts (one per lane), a few per-eigenvalue bisection rounds and
Rayleigh-quotient iteration on the tridiagonal matrix (one kept eigenvalue per lane).  Mirrors csrc/wave_trid.h step by step so that
the kernel can be checked against it; build-container diagnostic, not product code."""
import numpy as np


def householder_tridiag(A, b):
    """returns d, e, reflectors (list of (v, beta)), bhat = Qh' b"""
    A = A.copy(); b = b.copy()
    n = A.shape[0]
    refl = []
    for k in range(n - 2):
        x = A[k + 1:, k].copy()
        tail = np.sum(x[1:] ** 2)
        if tail == 0.0:
            refl.append(None)
            continue
        sigma = tail + x[0] ** 2
        nrm = np.sqrt(sigma)
        alpha = -nrm if x[0] > 0 else nrm
        v = x.copy(); v[0] -= alpha
        beta = 1.0 / (sigma + abs(x[0]) * nrm)       # 2 / v'v
        sub = A[k + 1:, k + 1:]
        p = beta * (sub @ v)
        K = 0.5 * beta * (p @ v)
        q = p - K * v
        sub -= np.outer(v, q) + np.outer(q, v)
        A[k + 1, k] = A[k, k + 1] = alpha
        A[k + 2:, k] = 0.0; A[k, k + 2:] = 0.0
        b[k + 1:] -= beta * v * (v @ b[k + 1:])
        refl.append((v, beta))
    return np.diag(A).copy(), np.diag(A, 1).copy(), refl, b


def sturm_count(d, e2, sig):
    """number of eigenvalues < sig (vectorised over sig)"""
    sig = np.atleast_1d(sig)
    n = d.shape[0]
    q = d[0] - sig
    cnt = (q < 0).astype(int)
    tiny = 1e-300
    for j in range(1, n):
        q = np.where(np.abs(q) < tiny, -tiny, q)
        q = d[j] - sig - e2[j - 1] / q
        cnt += (q < 0)
    return cnt


def tri_solve_gepp(d, e, sig, rhs):
    """(T - sig I) y = rhs by Gaussian elimination with partial pivoting, vectorised over the leading axis (one system per lane).
    d, e shared; sig (L,), rhs (L, n).  Returns y (L, n)."""
    L, n = rhs.shape
    # rows: (a_j, b_j, c_j) = current pivot row entries at columns j, j+1, j+2
    u0 = np.zeros((L, n)); u1 = np.zeros((L, n)); u2 = np.zeros((L, n))
    r = rhs.copy()
    pa = np.broadcast_to(d[0], (L,)) - sig                       # current row: diag
    pb = np.full(L, e[0] if n > 1 else 0.0)                      # current row: super
    pc = np.zeros(L)
    pr = r[:, 0].copy()
    eps_piv = 1e-300
    for j in range(n - 1):
        # next row j+1: (e_j, d_{j+1}-sig, e_{j+1})
        na = np.full(L, e[j]); nb = d[j + 1] - sig; nc = np.full(L, e[j + 1] if j + 2 < n else 0.0)
        nr = r[:, j + 1].copy()
        swap = np.abs(na) > np.abs(pa)
        a1 = np.where(swap, na, pa); b1 = np.where(swap, nb, pb); c1 = np.where(swap, nc, pc); r1 = np.where(swap, nr, pr)
        a2 = np.where(swap, pa, na); b2 = np.where(swap, pb, nb); c2 = np.where(swap, pc, nc); r2 = np.where(swap, pr, nr)
        a1 = np.where(a1 == 0.0, eps_piv, a1)
        m = a2 / a1
        u0[:, j] = a1; u1[:, j] = b1; u2[:, j] = c1; r[:, j] = r1
        pa = b2 - m * b1; pb = c2 - m * c1; pc = np.zeros(L); pr = r2 - m * r1
    pa = np.where(pa == 0.0, eps_piv, pa)
    u0[:, n - 1] = pa; r[:, n - 1] = pr
    y = np.zeros((L, n))
    y[:, n - 1] = r[:, n - 1] / u0[:, n - 1]
    if n > 1:
        y[:, n - 2] = (r[:, n - 2] - u1[:, n - 2] * y[:, n - 1]) / u0[:, n - 2]
    for j in range(n - 3, -1, -1):
        y[:, j] = (r[:, j] - u1[:, j] * y[:, j + 1] - u2[:, j] * y[:, j + 2]) / u0[:, j]
    return y


def trid_pinv_solve(d, e, bhat, tol, lanes=64, bis_rounds=3, rqi_max=12, stats=None):
    n = d.shape[0]
    e2 = e * e
    rad = np.zeros(n); rad[:-1] += np.abs(e); rad[1:] += np.abs(e)
    gl, gu = (d - rad).min(), (d + rad).max()
    nrmT = max(abs(gl), abs(gu))
    c_neg = int(sturm_count(d, e2, -tol)[0])                     # eigenvalues < -tol
    c_pos = int(sturm_count(d, e2, np.nextafter(tol, np.inf))[0])  # eigenvalues <= tol
    m_neg, m_pos = c_neg, n - c_pos
    kept = m_neg + m_pos
    if kept == 0:
        return np.zeros(n), 0
    # geometric shifts: positives on (tol, gu], negatives on [gl, -tol); lanes split in proportion to the counts
    lo = np.zeros(kept); hi = np.zeros(kept); idx = np.zeros(kept, dtype=int)
    def isolate(a, b, k0, m, L):
        """brackets of the eigenvalues number k0 .. k0+m-1 (0-based, ascending) inside (a, b), a, b > 0 as magnitudes; sign handled by caller"""
        pass
    # positive side
    out_lo, out_hi, out_k = [], [], []
    for side, m, k0 in (("neg", m_neg, 0), ("pos", m_pos, c_pos)):
        if m == 0:
            continue
        Ls = max(2, int(round(lanes * m / kept)))
        top = (gu if side == "pos" else -gl) * (1 + 1e-12) + 1e-300
        ratio = (top / tol) ** (1.0 / (Ls - 1))
        mags = tol * ratio ** np.arange(Ls)                       # tol .. top
        sh = mags if side == "pos" else -mags
        cnt = sturm_count(d, e2, sh)
        for k in range(k0, k0 + m):
            if side == "pos":
                # smallest shift index with count > k
                j = int(np.argmax(cnt > k))
                a, b = (mags[j - 1] if j > 0 else tol), mags[j]
                out_lo.append(a); out_hi.append(b)
            else:
                # negative: shifts descending in value as index grows; count(sh) <= k means eigenvalue k is >= sh
                j = int(np.argmax(cnt <= k))
                a, b = -mags[j], (-mags[j - 1] if j > 0 else -tol)
                out_lo.append(a); out_hi.append(b)
            out_k.append(k)
    lo = np.array(out_lo); hi = np.array(out_hi); idx = np.array(out_k)
    rounds = 0
    while True:                                                    # bisection until every bracket holds exactly one eigenvalue
        mid = 0.5 * (lo + hi)
        cnt = sturm_count(d, e2, mid)
        right = cnt <= idx                                         # eigenvalue idx is >= mid
        lo = np.where(right, mid, lo); hi = np.where(right, hi, mid)
        rounds += 1
        iso = (sturm_count(d, e2, hi) - sturm_count(d, e2, lo)) == 1   # (kernel: counts carried along, no extra evaluations)
        if (rounds >= bis_rounds and iso.all()) or rounds >= 60:
            break
    sig = 0.5 * (lo + hi)
    L = kept
    y = np.ones((L, n)) / np.sqrt(n)
    y = y * (1.0 + 0.37 * np.cos(np.outer(np.arange(L) + 1.0, np.arange(n) + 1.0)))
    y /= np.linalg.norm(y, axis=1, keepdims=True)
    done = np.zeros(L, dtype=bool)
    its = 0
    for its in range(1, rqi_max + 1):
        cnt = sturm_count(d, e2, sig)
        right = cnt <= idx
        lo = np.where(right & ~done, sig, lo); hi = np.where(~right & ~done, sig, hi)
        z = tri_solve_gepp(d, e, sig, y)
        nz = np.linalg.norm(z, axis=1, keepdims=True)
        z = z / nz
        Tz = d * z
        Tz[:, :-1] += e * z[:, 1:]
        Tz[:, 1:] += e * z[:, :-1]
        rho = np.sum(z * Tz, axis=1)
        res = np.linalg.norm(Tz - rho[:, None] * z, axis=1)
        inside = (rho >= lo) & (rho <= hi)
        newsig = np.where(inside, rho, 0.5 * (lo + hi))
        y = np.where(done[:, None], y, z)
        lamv = np.where(done, sig, rho) if its > 1 else rho
        done_new = done | (inside & (res <= 1e-15 * nrmT))
        sig = np.where(done, sig, np.where(done_new, rho, newsig))
        done = done_new
        if done.all():
            break
    if stats is not None:
        stats.append((rounds, its, int((~done).sum())))
    lam = sig
    coef = (y @ bhat) / lam
    return coef @ y, kept


def pinv_solve_sym(S, b, tol, stats=None):
    d, e, refl, bhat = householder_tridiag(S, b)
    xh, kept = trid_pinv_solve(d, e, bhat, tol, stats=stats)
    n = S.shape[0]
    x = xh.copy()
    for k in range(n - 3, -1, -1):
        if refl[k] is None:
            continue
        v, beta = refl[k]
        x[k + 1:] -= beta * v * (v @ x[k + 1:])
    return x, kept


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    worst = 0
    for trial in range(200):
        n = 31
        Hh = rng.standard_normal((19, 19)); Hh = Hh @ Hh.T * rng.uniform(0.1, 30) + np.diag(rng.uniform(0, 500, 19))
        Cc = rng.standard_normal((12, 19)) * 0.02
        S = np.block([[Hh, Cc.T], [Cc, -1e-6 * np.eye(12)]])
        b = rng.standard_normal(n)
        tol = 39 * np.spacing(10 ** rng.uniform(11, 14.5))
        st = []
        x, kept = pinv_solve_sym(S, b, tol, st)
        lam, V = np.linalg.eigh(S)
        k = np.abs(lam) > tol
        xr = V[:, k] @ ((V[:, k].T @ b) / lam[k])
        err = np.linalg.norm(x - xr) / np.linalg.norm(xr)
        worst = max(worst, err)
        if err > 1e-10 or kept != k.sum():
            print("trial", trial, "err %.2e" % err, kept, k.sum(), st)
    print("worst rel err %.2e" % worst)

def debug_trial(trial_want):
    rng = np.random.default_rng(0)
    for trial in range(trial_want + 1):
        n = 31
        Hh = rng.standard_normal((19, 19)); Hh = Hh @ Hh.T * rng.uniform(0.1, 30) + np.diag(rng.uniform(0, 500, 19))
        Cc = rng.standard_normal((12, 19)) * 0.02
        S = np.block([[Hh, Cc.T], [Cc, -1e-6 * np.eye(12)]])
        b = rng.standard_normal(n)
        tol = 39 * np.spacing(10 ** rng.uniform(11, 14.5))
    return S, b, tol

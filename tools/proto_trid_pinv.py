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


def twisted_rqi(d, e, lo, hi, idx, nrmT, res_tol=2e-15, maxit=40, stats=None):
    """One eigenpair of the tridiagonal matrix per lane.  [lo, hi] holds exactly eigenvalue number idx (ascending, 0-based).
    Rayleigh-quotient iteration with the twisted factorisation N_r D N_r' = T - sigma I as the solver (Parlett-Dhillon `getvec`):
    forward pivots Dp (their signs give the inertia, i.e. a free Sturm count that shrinks the bracket), backward pivots Dm,
    gamma_k = Dp_k + Dm_k - (d_k - sigma), twist at r = argmin |gamma|, z_r = 1, z_i = -(e_i / Dp_i) z_{i+1} (i < r),
    z_{i+1} = -(e_i / Dm_{i+1}) z_i (i >= r); (T - sigma) z = gamma_r e_r, so the Rayleigh-quotient correction is gamma_r / |z|^2
    and the residual |gamma_r| / |z|.  A correction that leaves the bracket is replaced by a bisection step."""
    L = lo.shape[0]; n = d.shape[0]
    sig = 0.5 * (lo + hi)
    done = np.zeros(L, dtype=bool)
    zfin = np.zeros((L, n)); lam = sig.copy()
    ar = np.arange(L)
    it = 0
    for it in range(1, maxit + 1):
        Dp = np.zeros((L, n)); Dm = np.zeros((L, n))
        Dp[:, 0] = d[0] - sig
        for i in range(n - 1):
            q = np.where(np.abs(Dp[:, i]) < 1e-300, -1e-300, Dp[:, i])
            Dp[:, i] = q
            Dp[:, i + 1] = (d[i + 1] - sig) - e[i] * e[i] / q
        neg = np.sum(Dp < 0, axis=1)
        Dm[:, n - 1] = d[n - 1] - sig
        for i in range(n - 2, -1, -1):
            q = np.where(np.abs(Dm[:, i + 1]) < 1e-300, -1e-300, Dm[:, i + 1])
            Dm[:, i + 1] = q
            Dm[:, i] = (d[i] - sig) - e[i] * e[i] / q
        gam = Dp + Dm - (d[None, :] - sig[:, None])
        r = np.argmin(np.abs(gam), axis=1)
        z = np.zeros((L, n))
        z[ar, r] = 1.0
        for i in range(n - 2, -1, -1):
            z[:, i] = np.where(i < r, -(e[i] / Dp[:, i]) * z[:, i + 1], z[:, i])
        for i in range(n - 1):
            z[:, i + 1] = np.where(i >= r, -(e[i] / Dm[:, i + 1]) * z[:, i], z[:, i + 1])
        nz2 = np.sum(z * z, axis=1)
        gr = gam[ar, r]
        corr = gr / nz2
        resid = np.abs(gr) / np.sqrt(nz2)
        right = neg <= idx                                         # eigenvalue idx is >= sigma
        lo = np.where(right & ~done, sig, lo); hi = np.where(~right & ~done, sig, hi)
        conv = (resid <= res_tol * nrmT) | (np.abs(corr) <= 4e-16 * np.abs(sig))
        newly = conv & ~done
        zfin[newly] = z[newly]; lam[newly] = (sig + corr)[newly]
        done |= conv
        cand = sig + corr
        inside = (cand >= lo) & (cand <= hi)
        sig = np.where(done, sig, np.where(inside, cand, 0.5 * (lo + hi)))
        if done.all():
            break
    if stats is not None:
        stats.append((it, int((~done).sum())))
    zfin[~done] = z[~done]
    return lam, zfin


def trid_pinv_solve(d, e, bhat, tol, lanes=64, bis_rounds=2, res_tol=2e-15, stats=None):
    """x = sum over the eigenvalues |lambda| > tol of the tridiagonal matrix of y (y' bhat) / lambda; returns x and the kept count"""
    n = d.shape[0]; e2 = e * e
    rad = np.zeros(n); rad[:-1] += np.abs(e); rad[1:] += np.abs(e)
    gl, gu = (d - rad).min(), (d + rad).max()
    nrmT = max(abs(gl), abs(gu))
    c_neg = int(sturm_count(d, e2, -tol)[0])                       # eigenvalues < -tol
    c_pos = int(sturm_count(d, e2, np.nextafter(tol, np.inf))[0])  # eigenvalues <= tol
    m_neg, m_pos = c_neg, n - c_pos
    kept = m_neg + m_pos
    if kept == 0:
        return np.zeros(n), 0
    out_lo, out_hi, out_k = [], [], []
    for side, m, k0 in (("neg", m_neg, 0), ("pos", m_pos, c_pos)):  # one Sturm count per lane on geometrically spaced shifts
        if m == 0:
            continue
        Ls = max(2, int(round(lanes * m / kept)))
        top = (gu if side == "pos" else -gl) * (1 + 1e-12) + 1e-300
        ratio = (top / tol) ** (1.0 / (Ls - 1))
        mags = tol * ratio ** np.arange(Ls)
        cnt = sturm_count(d, e2, mags if side == "pos" else -mags)
        for k in range(k0, k0 + m):
            if side == "pos":
                j = int(np.argmax(cnt > k)); a, b = (mags[j - 1] if j > 0 else tol), mags[j]
            else:
                j = int(np.argmax(cnt <= k)); a, b = -mags[j], (-mags[j - 1] if j > 0 else -tol)
            out_lo.append(a); out_hi.append(b); out_k.append(k)
    lo = np.array(out_lo); hi = np.array(out_hi); idx = np.array(out_k)
    clo = sturm_count(d, e2, lo); chi = sturm_count(d, e2, hi)
    rounds = 0
    while True:                                                    # bisection until every bracket holds exactly one eigenvalue
        iso = (chi - clo) == 1
        if (rounds >= bis_rounds and iso.all()) or rounds >= 60:
            break
        mid = 0.5 * (lo + hi); cnt = sturm_count(d, e2, mid)
        right = cnt <= idx
        lo = np.where(right, mid, lo); clo = np.where(right, cnt, clo)
        hi = np.where(right, hi, mid); chi = np.where(right, chi, cnt)
        rounds += 1
    st = []
    lam, z = twisted_rqi(d, e, lo, hi, idx, nrmT, res_tol=res_tol, stats=st)
    if stats is not None:
        stats.append((rounds,) + st[0])
    coef = (z @ bhat) / (lam * np.sum(z * z, axis=1))
    return coef @ z, kept


def pinv_solve_sym(S, b, tol, stats=None, **kw):
    d, e, refl, bhat = householder_tridiag(S, b)
    xh, kept = trid_pinv_solve(d, e, bhat, tol, stats=stats, **kw)
    n = S.shape[0]
    x = xh.copy()
    for k in range(n - 3, -1, -1):
        if refl[k] is None:
            continue
        v, beta = refl[k]
        x[k + 1:] -= beta * v * (v @ x[k + 1:])
    return x, kept


def random_kkt(rng, n1=19, n2=12):
    Hh = rng.standard_normal((n1, n1)); Hh = Hh @ Hh.T * rng.uniform(0.1, 30) + np.diag(rng.uniform(0, 500, n1))
    Cc = rng.standard_normal((n2, n1)) * 0.02
    S = np.block([[Hh, Cc.T], [Cc, -1e-6 * np.eye(n2)]])
    return S, rng.standard_normal(n1 + n2), 39 * np.spacing(10 ** rng.uniform(11, 14.5))


if __name__ == "__main__":
    import sys
    kw = dict(bis_rounds=int(sys.argv[1]) if len(sys.argv) > 1 else 2, res_tol=float(sys.argv[2]) if len(sys.argv) > 2 else 2e-15)
    rng = np.random.default_rng(0)
    worst, allst = 0, []
    for trial in range(400):
        S, b, tol = random_kkt(rng)
        st = []
        x, kept = pinv_solve_sym(S, b, tol, st, **kw)
        allst += st
        lam, V = np.linalg.eigh(S)
        k = np.abs(lam) > tol
        xr = V[:, k] @ ((V[:, k].T @ b) / lam[k])
        err = np.linalg.norm(x - xr) / np.linalg.norm(xr)
        worst = max(worst, err)
        if err > 1e-10 or kept != k.sum():
            print("trial", trial, "err %.2e" % err, kept, k.sum(), st)
    a = np.array(allst)
    print("worst rel err %.2e; bisection rounds mean %.1f max %d; RQI iterations mean %.1f max %d; not converged %d"
          % (worst, a[:, 0].mean(), a[:, 0].max(), a[:, 1].mean(), a[:, 1].max(), a[:, 2].sum()))

"""Prototype (numpy fp64) of the factored Gauss-Helmert step for Faugeras-Papadopoulo's parameterisation, measured against the
50-digit fixture tests/golden/gh_mp_faugpapa.npz.  Build-container diagnostic: it prototypes the arithmetic the HIP kernel
(csrc/gh_wg_kernel.h, FaugPapaModel) uses -- strong directions of the weight blocks kept as factors, an orthogonal change of basis
that aligns the 1e12-weighted subspace of A'WA with coordinate axes, block elimination of that subspace, truncated pseudo-inverse of the
30 x 30 remainder -- nothing here is product code.
Usage: python tools/proto_faugpapa_factored.py [case [scenes]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from oracle import tft_oracle as O
from oracle import gh_mp_oracle as G
from helpers import rel_err_T, rel_err, golden_cases
import proto_trid_pinv as TP

EPS12 = 1e-12
VERBOSE = int(os.environ.get("PROTO_VERBOSE", "0"))


def blocks(xi, T):
    """per-correspondence f (N,4), Ap (N,4,27), B (N,4,6) -- FaugPapaTFTPoseEstimation.m:96-112"""
    N = xi.shape[0] // 6
    f, Ap, B = O._trilinear_blocks(xi, T)
    Bb = np.stack([B[4 * i:4 * i + 4, 6 * i:6 * i + 6] for i in range(N)])
    return f.reshape(N, 4), Ap.reshape(N, 4, 27), Bb


def strong_basis(Hs, rel=1e-7, absmin=1e7):
    """orthogonal Q (27 x 27) whose first ns columns span the dominant columns of the strong Gram matrix: diagonally pivoted Cholesky
    until the pivot falls under max(rel * first pivot, absmin), Householder QR of the factor, Q = H_1 ... H_ns"""
    n = Hs.shape[0]
    dg = np.diag(Hs).copy()
    L = np.zeros((n, 0))
    used = np.zeros(n, dtype=bool)
    d0 = None
    while L.shape[1] < n:
        p = int(np.argmax(np.where(used, -1.0, dg)))
        if d0 is None:
            d0 = dg[p]
        if not (dg[p] > max(rel * d0, absmin)):
            break
        col = Hs[:, p] - L @ L[p, :]
        col[used] = 0.0
        l = col / np.sqrt(dg[p])
        L = np.hstack([L, l[:, None]])
        dg = dg - l * l
        used[p] = True
    ns = L.shape[1]
    Q = np.eye(n)
    Lw = L.copy()
    for k in range(ns):
        x = Lw[k:, k].copy()
        nrm = np.linalg.norm(x)
        alpha = -nrm if x[0] > 0 else nrm
        v = x.copy(); v[0] -= alpha
        vv = v @ v
        if vv == 0.0:
            continue
        beta = 2.0 / vv
        Lw[k:, k:] -= beta * np.outer(v, v @ Lw[k:, k:])
        Q[:, k:] -= beta * np.outer(Q[:, k:] @ v, v)
    return Q, ns


def gh_factored(x, x_est, p0, it_max=400, mode="factored"):
    N = x.shape[0] // 6
    u, c = 27, 12
    xi, ti = x_est.copy(), p0.copy()
    objFunc = float((xi - x) @ (xi - x))
    reason, it = "itmax", 0
    for it in range(1, it_max + 1):
        T = O._unvecT(ti)
        _, g, _, _, C, _ = O._faugpapa_constrGH(xi[:6], ti)       # g, C only depend on ti
        f, Ap, B = blocks(xi, T)
        Wb = np.einsum("nij,nkj->nik", B, B) + EPS12 * np.eye(4)
        lam, V = np.linalg.eigh(Wb)
        lam_max = lam.max()
        tolW = 4 * N * np.spacing(lam_max)
        n = V[:, :, 0]                                             # strong direction per correspondence
        Btn = np.einsum("nij,ni->nj", B, n)
        mu = np.sum(Btn * Btn, axis=1)                             # |B'n|^2: no cancellation
        lam0 = mu + EPS12
        keep0 = lam0 > tolW
        cs = np.where(keep0, 1.0 / lam0, 0.0)
        K = np.zeros((N, 4, 4))
        for k in range(1, 4):
            kk = lam[:, k] > tolW
            K += np.where(kk, 1.0 / lam[:, k], 0.0)[:, None, None] * np.einsum("ni,nj->nij", V[:, :, k], V[:, :, k])
        K += EPS12 * np.eye(4)
        # the +1e-12 I of Gauss_Helmert.m:57 also adds 1e-12 along n: negligible beside cs but kept (it is in K through the identity)
        xd = (x - xi).reshape(N, 6)
        w = -f - np.einsum("nij,nj->ni", B, xd)
        om = -np.einsum("ni,ni->n", n, f) - np.einsum("nj,nj->n", Btn, xd)      # n'w
        KA = np.einsum("nij,njk->nik", K, Ap)
        R = np.einsum("nji,njk->ik", Ap, KA)
        r = np.einsum("nji,nj->i", KA, w)
        a = np.einsum("nji,nj->ni", Ap, n)                         # N x 27
        sq = np.sqrt(cs)
        if mode == "plain":
            H = R + (a * cs[:, None]).T @ a
            bb = r + a.T @ (cs * om)
            M = np.block([[H, C.T], [C, np.zeros((c, c))]]) + EPS12 * np.eye(u + c)
            lamM, VM = np.linalg.eigh(M)
            tol = (u + c) * np.spacing(np.abs(lamM).max())
            kept = np.abs(lamM) > tol
            b = np.concatenate([bb, -g])
            aux = VM[:, kept] @ ((VM[:, kept].T @ b) / lamM[kept])
            dt = aux[:u]
            rho = a @ dt - om
        else:
            if mode == "kernel":
                Hs = (a * cs[:, None]).T @ a                      # strong Gram alone, fp64
                Q, ns = strong_basis(Hs)
                lamH = np.zeros(27)
            else:
                Hf = R + (a * cs[:, None]).T @ a
                lamH, Q = np.linalg.eigh(Hf)
                lamH, Q = lamH[::-1], Q[:, ::-1]
                ns = int(np.sum(lamH > 1e-7 * lamH[0]))
            if VERBOSE:
                print("  it %d  eig(H) %s  ns %d  cs range %.1e..%.1e" % (it, np.array2string(lamH, precision=1, max_line_width=400), ns, cs.min(), cs.max()))
            ap = a @ Q                                             # N x 27, tangential columns tiny
            Gm = ap * sq[:, None]
            Hp = Gm.T @ Gm + Q.T @ R @ Q
            bp = Gm.T @ (sq * om) + Q.T @ r
            Cp = C @ Q
            Mp = np.block([[Hp, Cp.T], [Cp, np.zeros((c, c))]]) + EPS12 * np.eye(u + c)
            b = np.concatenate([bp, -g])
            # ||M||_2 for pinv's tolerance: largest eigenvalue of the strong block (Schur corrections are far below its spacing)
            nrm = np.linalg.eigvalsh(Mp[:ns, :ns]).max() if ns else None
            M11 = Mp[:ns, :ns]; M12 = Mp[:ns, ns:]; M22 = Mp[ns:, ns:]
            L = np.linalg.cholesky(M11) if ns else None
            Y = np.linalg.solve(M11, M12) if ns else np.zeros((0, u + c))
            Sg = M22 - M12.T @ Y
            Sg = 0.5 * (Sg + Sg.T)
            b2 = b[ns:] - Y.T @ b[:ns]
            lamS, VS = np.linalg.eigh(Sg)
            if nrm is None:
                nrm = np.abs(lamS).max()
            tol = (u + c) * np.spacing(nrm)
            kept = np.abs(lamS) > tol
            if mode == "kernel":
                z2, nk = TP.pinv_solve_sym(Sg, b2, tol)
                assert nk == kept.sum(), (nk, kept.sum())
            else:
                z2 = VS[:, kept] @ ((VS[:, kept].T @ b2) / lamS[kept])
            z1 = np.linalg.solve(M11, b[:ns] - M12 @ z2) if ns else np.zeros(0)
            z = np.concatenate([z1, z2])
            dt = Q @ z[:u]
            rho = (a @ dt - om) if os.environ.get("PROTO_RHO_ORIG") else (ap @ z[:u] - om)
            if VERBOSE:
                print("      kept %d of %d in the remainder (tol %.3g); |dt| %.3e" % (kept.sum(), len(lamS), tol, np.linalg.norm(dt)))
        res = np.einsum("nij,j->ni", Ap, dt) - w
        v = -(np.einsum("nji,nj->ni", B, np.einsum("nij,nj->ni", K, res)) + Btn * (cs * rho)[:, None]).reshape(6 * N)
        if np.linalg.norm(dt) < 1e-6 and np.linalg.norm(xi - x - v) < 1e-6:
            reason = "converged"; break
        obj = float(v @ v)
        if obj > objFunc:
            reason = "rose"; break
        objFunc = obj
        xi = x + v
        ti = ti + dt
    return ti, xi, it, reason


def run(case=None, scenes=None, mode="factored"):
    g = np.load(os.path.join(ROOT, "tests", "golden", "gh_mp_faugpapa.npz"))
    for ci, pre in golden_cases(g):
        if case is not None and ci != case:
            continue
        N, B, noise = g[pre + "meta"]
        N, B = int(N), int(B)
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        devs, dits = [], []
        for b in range(B if scenes is None else min(B, scenes)):
            Cb = C[b].T.copy()
            x, x_est, p0, normals = G.faugpapa_start(Cb, CalM)
            p, _, it, reason = gh_factored(x, x_est, p0, mode=mode)
            T = O.transform_TFT(O._unvecT(p), normals[0], normals[1], normals[2], 1)
            R2, R3 = O.R_t_from_TFT(T, CalM, Cb)
            d = max(rel_err_T(T, g[pre + "mp_T"][b]), rel_err(R2, g[pre + "mp_Rt2"][b]), rel_err(R3, g[pre + "mp_Rt3"][b]))
            devs.append(d); dits.append(it - int(g[pre + "mp_iter"][b]))
            if VERBOSE:
                print("N=%d scene %d: dev %.2e it %d (mp %d) %s" % (N, b, d, it, int(g[pre + "mp_iter"][b]), reason))
        devs = np.array(devs)
        print("N=%-4d %s: p50 %.1e p90 %.1e max %.1e   iter diff %s" % (N, mode, np.quantile(devs, 0.5), np.quantile(devs, 0.9), devs.max(), np.bincount(np.abs(dits)).tolist()))


if __name__ == "__main__":
    case = int(sys.argv[1]) if len(sys.argv) > 1 else None
    scenes = int(sys.argv[2]) if len(sys.argv) > 2 else None
    for mode in os.environ.get("PROTO_MODES", "plain,factored").split(","):
        run(case, scenes, mode)

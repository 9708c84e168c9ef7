"""
TEST INFRASTRUCTURE ONLY -- block-structured restatement of Gauss_Helmert.m for the
trilinearity-based methods (Ressl first), in numpy.

Optimization/Gauss_Helmert.m builds dense 4N x 6N / 4N x 4N matrices and calls pinv on
them; their structure is block-diagonal (one 4 x 6 block B_i per correspondence), so
  W = B pinv(P) B'                 -> 4x4 blocks  W_i = B_i B_i'
  pinv(W + 1e-12 I) + 1e-12 I      -> per-block eigen-decomposition with the GLOBAL
                                      tolerance 4N * eps(max_i lambda_max(W_i + 1e-12 I))
  A'WA = D' (sum_i Ap_i' W_i Ap_i) D  (A_i = Ap_i D, D = dT/dparams)
  pinv(M + 1e-12 I) b               -> direct solve when no singular value falls under
                                      pinv's tolerance (checked), which is the generic case.
This is the algebra the HIP kernels use.  It agrees with the dense restatement
(oracle/tft_oracle.py) only to ~1e-5 in the parameters: the reference's exit test
"objective rose" (Gauss_Helmert.m:75) compares objectives that differ by ~1e-9
relative once the iteration has stagnated, so which iteration stops -- and whether the
last ~1e-5 step is applied -- differs between any two floating-point evaluations
(hard part 1).  PARITY UNPINNED, as for the whole oracle.
"""
import numpy as np

from oracle import tft_oracle as O


def eps_of(x):
    return np.spacing(abs(x))


def ressl_model(p, Ind):
    """ResslTFTPoseEstimation.m:112-135,164-170: params -> T, D = dT/dp (27x20), g, C."""
    Ind2 = [k for k in range(3) if k != Ind]
    S = p[0:9].reshape(3, 3, order='F')
    e21 = np.ones(3); e21[Ind2] = p[9:11]
    e31 = p[17:20]
    mn = np.zeros((3, 3)); mn[:, Ind2] = p[11:17].reshape(3, 2, order='F')
    T = np.zeros((3, 3, 3)); D = np.zeros((27, 20))
    for i in range(3):
        for k in range(3):
            for j in range(3):
                r = j + 3 * k + 9 * i
                T[j, k, i] = e21[j] * S[k, i] + mn[i, j] * e31[k]
                D[r, k + 3 * i] = e21[j]
                for m in range(2):
                    if j == Ind2[m]:
                        D[r, 9 + m] = S[k, i]
                        D[r, 11 + i + 3 * m] = e31[k]
                D[r, 17 + k] = mn[i, j]
    g = np.array([np.sum(e31 ** 2) - 1, np.sum(S ** 2) - 1])
    C = np.zeros((2, 20)); C[0, 17:20] = 2 * e31; C[1, 0:9] = 2 * S.reshape(9, order='F')
    return T, D, g, C


def point_blocks(xi6, T):
    """The per-correspondence block of ResslTFTPoseEstimation.m:141-161."""
    x1, y1, x2, y2, x3, y3 = xi6
    h = np.array([x1, y1, 1.0])
    S2 = np.array([[0, -1.], [-1, 0], [y2, x2]]); S3 = np.array([[0, -1.], [-1, 0], [y3, x3]])
    M = h[0] * T[:, :, 0] + h[1] * T[:, :, 1] + h[2] * T[:, :, 2]
    f = (S2.T @ M @ S3).reshape(4, order='F')
    B = np.zeros((4, 6))
    B[:, 0] = (S2.T @ T[:, :, 0] @ S3).reshape(4, order='F')
    B[:, 1] = (S2.T @ T[:, :, 1] @ S3).reshape(4, order='F')
    u3 = S3.T @ M[2, :]; u2 = S2.T @ M[:, 2]
    for a3 in range(2):
        for a2 in range(2):
            r = a2 + 2 * a3
            B[r, 2] = u3[a3] * (a2 == 1); B[r, 3] = u3[a3] * (a2 == 0)
            B[r, 4] = u2[a2] * (a3 == 1); B[r, 5] = u2[a2] * (a3 == 0)
    K = np.kron(S3, S2)
    return f, B, K, h


def gauss_helmert_blocks(model, x, xi0, p0, N, it_max=400, tol=1e-6):
    xi = xi0.copy(); p = p0.copy(); u = p.size
    obj = float((xi0 - x) @ (xi0 - x)); reason = 'itmax'; it = 0
    for it in range(1, it_max + 1):
        T, D, g, C = model(p); c = C.shape[0]
        eig = []; fs = []; Bs = []; Ks = []; hs = []; smax = 0.0
        for i in range(N):
            f, B, K, h = point_blocks(xi[6 * i:6 * i + 6], T)
            lam, V = np.linalg.eigh(B @ B.T + 1e-12 * np.eye(4))
            smax = max(smax, np.abs(lam).max())
            eig.append((lam, V)); fs.append(f); Bs.append(B); Ks.append(K); hs.append(h)
        if not np.isfinite(smax):
            reason = 'nanW'; break
        tolW = 4 * N * eps_of(smax)
        G = np.zeros((27, 27)); gv = np.zeros(27); Wp = []; ww = []
        for i in range(N):
            lam, V = eig[i]; keep = lam > tolW
            Wi = (V[:, keep] / lam[keep]) @ V[:, keep].T + 1e-12 * np.eye(4)
            w = -fs[i] - Bs[i] @ (x[6 * i:6 * i + 6] - xi[6 * i:6 * i + 6])
            Ap = np.kron(hs[i].reshape(1, 3), Ks[i].T)
            G += Ap.T @ Wi @ Ap; gv += Ap.T @ (Wi @ w); Wp.append(Wi); ww.append(Wi @ w)
        Nm = D.T @ G @ D; r = D.T @ gv
        M = np.block([[Nm, C.T], [C, np.zeros((c, c))]]) + 1e-12 * np.eye(u + c)
        b = np.concatenate([r, -g])
        if not np.all(np.isfinite(M)):
            reason = 'nanM'; break
        sv = np.linalg.svd(M, compute_uv=False)
        if not (sv > (u + c) * eps_of(sv[0])).all():
            raise NotImplementedError("rank-deficient KKT matrix: pinv truncation path")
        dt = np.linalg.solve(M, b)[:u]
        dT = D @ dt
        v = np.zeros(6 * N)
        for i in range(N):
            Ap = np.kron(hs[i].reshape(1, 3), Ks[i].T)
            v[6 * i:6 * i + 6] = -Bs[i].T @ (Wp[i] @ (Ap @ dT) - ww[i])
        if np.linalg.norm(dt) < tol and np.linalg.norm(xi - x - v) < tol:
            reason = 'converged'; break
        o = float(v @ v)
        if o > obj:
            reason = 'rose'; break
        obj = o; xi = x + v; p = p + dt
    return xi, p, it, reason


def ressl_setup(Corresp):
    """ResslTFTPoseEstimation.m:48-82: normalisation, linearTFT, initial parameters and observations."""
    x1, N1 = O.Normalize2Ddata(Corresp[0:2]); x2, N2 = O.Normalize2Ddata(Corresp[2:4]); x3, N3 = O.Normalize2Ddata(Corresp[4:6])
    T, P1, P2, P3 = O.linearTFT(x1, x2, x3)
    e21 = P2[:, 3].copy(); Ind = int(np.argmax(np.abs(e21))); e21 = e21 / e21[Ind]
    e31 = P3[:, 3] / np.linalg.norm(P3[:, 3])
    S = np.stack([T[Ind, :, 0], T[Ind, :, 1], T[Ind, :, 2]], axis=1); aux = np.linalg.norm(S); S = S / aux; T = T / aux
    Ind2 = [k for k in range(3) if k != Ind]
    mn = np.stack([e31 @ (T[:, :, i].T - np.outer(S[:, i], e21)) for i in range(3)], axis=0)[:, Ind2]
    x, x_est = O._gh_initial_obs(P1, P2, P3, x1, x2, x3)
    p = np.concatenate([S.reshape(9, order='F'), e21[Ind2], mn.reshape(6, order='F'), e31])
    return dict(Ind=Ind, p0=p, x=x, x_est=x_est, normals=(N1, N2, N3))


def ResslTFTPoseEstimation_blocks(Corresp, CalM, return_debug=False):
    """ResslTFTPoseEstimation.m:47-105 with the block-structured Gauss-Helmert."""
    s = ressl_setup(Corresp)
    N = Corresp.shape[1]
    xi, p_opt, it, reason = gauss_helmert_blocks(lambda q: ressl_model(q, s["Ind"]), s["x"], s["x_est"], s["p0"], N)
    T = ressl_model(p_opt, s["Ind"])[0]
    T = O.transform_TFT(T, *s["normals"], 1)
    R_t_2, R_t_3 = O.R_t_from_TFT(T, CalM, Corresp)
    Reconst = O._final_reconst(CalM, R_t_2, R_t_3, Corresp)
    if return_debug:
        return R_t_2, R_t_3, Reconst, T, it, dict(reason=reason, p0=s["p0"], p_opt=p_opt, Ind=s["Ind"])
    return R_t_2, R_t_3, Reconst, T, it

"""TEST INFRASTRUCTURE ONLY -- never imported by the product (tft_vs_fund_amd/).

Extended-precision (mpmath, 50 digits) evaluation of the reference's Gauss-Helmert iteration
(Optimization/Gauss_Helmert.m:38-83) with Ressl's callback (TFT_methods/ResslTFTPoseEstimation.m:110-177), Nordberg's
(TFT_methods/NordbergTFTPoseEstimation.m:128-222) Faugeras-Papadopoulo's (TFT_methods/FaugPapaTFTPoseEstimation.m:87-159) and the two Ponce-Hebert
Pi-matrix ones (TFT_methods/PiPoseEstimation.m:109-182, TFT_methods/PiColPoseEstimation.m:144-216).

Purpose: `pinv(W + 1e-12 I)` gives every correspondence one direction of weight ~1e12, so A'WA
cancels ten digits in ANY fp64 evaluation -- the reference's own dense MATLAB product included.  To judge the HIP kernel
one therefore needs the iteration the reference's FORMULAS define, free of fp64 rounding: this file.  Both the LAPACK-backed
numpy oracle (oracle/tft_oracle.py, a stand-in for MATLAB's arithmetic) and the kernel are then measured against it; the
kernel is required to be no noisier than the LAPACK evaluation (tests/test_gpu_gh_noise.py, profiles/r2_gh_noise_mp.txt).

What is exact here and what is not:
  * every sum, product, eigen-decomposition and the stop tests run in 50-digit arithmetic;
  * B (4N x 6N) and W = B B' (4N x 4N) are block diagonal (one 4 x 6 / 4 x 4 block per correspondence) -- an identity of
    the reference's formulas, not an approximation -- so pinv(W + 1e-12 I) is taken block by block;
  * MATLAB's pinv tolerance max(size(A)) * eps(norm(A)) (documented semantics) is a fixed fp64 number once norm(A) is
    known: eps() is the double-precision spacing of the (exactly computed, then rounded) 2-norm;
  * `1e-12` is the double nearest to 1e-12, as MATLAB parses it;
  * the START of the iteration (parameters p0 and observations x_est from linearTFT and the projective triangulation,
    ResslTFTPoseEstimation.m:47-81) and the tail after it (transform_TFT, R_t_from_TFT) are the fp64 numpy oracle's:
    they agree with the kernel's to ~1e-11 and are not what is being measured.
"""
import numpy as np
import mpmath as mp

from oracle import tft_oracle as O

mp.mp.dps = 50
_EPS12 = mp.mpf(float(1e-12))
_mpf = np.frompyfunc(mp.mpf, 1, 1)
_flt = np.frompyfunc(float, 1, 1)


def to_mp(a):
    return _mpf(np.asarray(a, dtype=np.float64))


def to_float(a):
    return _flt(a).astype(np.float64)


def _zeros(*shape):
    z = np.empty(shape, dtype=object)
    z.fill(mp.mpf(0))
    return z


def _eigsy(Mx):
    """eigenvalues (ascending) and eigenvectors (columns) of a symmetric object matrix"""
    n = Mx.shape[0]
    E, Q = mp.eigsy(mp.matrix(Mx.tolist()))
    lam = np.array([E[i] for i in range(n)], dtype=object)
    V = np.array([[Q[i, j] for j in range(n)] for i in range(n)], dtype=object)
    return lam, V


def _matlab_tol(size_max, norm2):
    """max(size(A)) * eps(norm(A)) with eps() the double-precision spacing"""
    return mp.mpf(size_max * float(np.spacing(float(norm2))))


def _pinv_sym(Mx, size_max=None, norm2=None):
    """pinv of a symmetric matrix through its eigen-decomposition (singular values = |eigenvalues|), MATLAB tolerance"""
    lam, V = _eigsy(Mx)
    nrm = max(abs(l) for l in lam) if norm2 is None else norm2
    tol = _matlab_tol(Mx.shape[0] if size_max is None else size_max, nrm)
    out = _zeros(*Mx.shape)
    kept = 0
    for k in range(len(lam)):
        if abs(lam[k]) > tol:
            kept += 1
            out = out + np.outer(V[:, k], V[:, k]) / lam[k]
    return out, kept


def _ressl_unpack(p, Ind):
    Ind2 = [k for k in range(3) if k != Ind]
    S = p[0:9].reshape(3, 3, order='F')
    e21 = np.array([mp.mpf(1)] * 3, dtype=object)
    e21[Ind2] = p[9:11]
    e31 = p[17:20]
    mn = _zeros(3, 3)
    mn[:, Ind2] = p[11:17].reshape(3, 2, order='F')
    T = _zeros(3, 3, 3)
    for i in range(3):
        T[:, :, i] = (np.outer(S[:, i], e21) + np.outer(e31, mn[i, :])).T          # ResslTFT...m:118-120
    return S, e21, e31, mn, T, Ind2


def _ressl_D(S, e21, e31, mn, Ind2):
    """ResslTFTPoseEstimation.m:164-170: dT/dp, 27 x 20"""
    D = _zeros(27, 20)
    one = mp.mpf(1)
    aux = _zeros(3, 2)
    aux[Ind2[0], 0] = one
    aux[Ind2[1], 1] = one
    for i in range(3):
        for k in range(3):
            for j in range(3):
                r = j + 3 * k + 9 * i                                              # T(j,k,i) = e21(j) S(k,i) + e31(k) mn(i,j)  (transposed slices)
                D[r, k + 3 * i] = e21[j]                                           # d/dS(k,i)
                for a in range(2):
                    D[r, 9 + a] = D[r, 9 + a] + S[k, i] * aux[j, a]                # d/de21(Ind2[a])
                    D[r, 11 + i + 3 * a] = e31[k] * aux[j, a]                      # d/dmn(i,Ind2[a])   (mn(:) column-major: 3a + i)
                D[r, 17 + k] = mn[i, j]                                            # d/de31(k)
    return D


def _blocks(xi_i, T):
    """per-correspondence f (4), Ap (4 x 27), B (4 x 6): ResslTFTPoseEstimation.m:141-161"""
    x1, y1, x2, y2, x3, y3 = xi_i
    zero, one = mp.mpf(0), mp.mpf(1)
    S2 = np.array([[zero, -one], [-one, zero], [y2, x2]], dtype=object)
    S3 = np.array([[zero, -one], [-one, zero], [y3, x3]], dtype=object)
    h1 = np.array([x1, y1, one], dtype=object)
    T1, T2, T3 = T[:, :, 0], T[:, :, 1], T[:, :, 2]
    f = S2.T.dot(x1 * T1 + y1 * T2 + T3).dot(S3).reshape(4, order='F')
    Ap = _zeros(4, 27)
    for b in range(2):
        for a in range(2):
            for i in range(3):
                for k in range(3):
                    for j in range(3):
                        Ap[2 * b + a, j + 3 * k + 9 * i] = h1[i] * S3[k, b] * S2[j, a]
    J3 = T[2, :, :].T                                                                # rows T_i(3,:)
    K3 = T[:, 2, :]                                                                  # cols T_i(:,3)
    B = _zeros(4, 6)
    B[:, 0] = S2.T.dot(T1).dot(S3).reshape(4, order='F')
    B[:, 1] = S2.T.dot(T2).dot(S3).reshape(4, order='F')
    u = S3.T.dot(J3.T).dot(h1)                                                       # 2
    v = S2.T.dot(K3).dot(h1)                                                         # 2
    sw = [[zero, one], [one, zero]]
    for r in range(2):
        for c in range(2):
            for q in range(2):
                B[2 * r + q, 2 + c] = u[r] * sw[q][c]                                # kron(u, sw)
                B[2 * q + r, 4 + c] = sw[q][c] * v[r]                                # kron(sw, v)
    return f, Ap, B


def gauss_helmert_mp(x, x_est, p0, model, u, c, it_max=400, return_history=False):
    """Gauss_Helmert.m:38-83 in extended precision for a trifocal-tensor callback, from the fp64 start (x, x_est, p0).
    model(t) -> (T 3x3x3, D = dT(:)/dt 27 x u, g (c), C (c x u)), all object arrays; the per-correspondence blocks f, Ap, B are the
    ones every TFT callback shares (`_blocks`; A = Ap D).  Or model(t) -> (point_fn, g, C) with point_fn(o) -> (f 4, A 4 x u, B 4 x 6)
    for callbacks with blocks of their own (the Pi-matrix method).  Returns p_opt (float64), xi (float64), it, reason [, history]."""
    N = x.shape[0] // 6
    xm, xi, ti = to_mp(x), to_mp(x_est), to_mp(p0)
    tol = mp.mpf(float(1e-6))
    v0 = xi - xm
    objFunc = sum(v * v for v in v0)
    reason, it, hist = 'itmax', 0, []
    for it in range(1, it_max + 1):
        ev = model(ti)
        if len(ev) == 4:                                                             # trifocal-tensor callbacks: (T, D, g, C)
            T, D, g, C = ev
            point_fn = lambda o: (lambda fAB: (fAB[0], fAB[1].dot(D), fAB[2]))(_blocks(o, T))
        else:                                                                        # callbacks with their own blocks: (point_fn, g, C)
            point_fn, g, C = ev
        blocks = []
        lam_max = mp.mpf(0)
        E = 4
        for i in range(N):
            f, A, B = point_fn(xi[6 * i:6 * i + 6])
            E = f.shape[0]                                                           # equations per correspondence (4; PiCol: 5)
            Wb = B.dot(B.T)                                                          # :52 (P = I)
            lam, V = _eigsy(Wb + _EPS12 * np.eye(E, dtype=object))
            lam_max = max(lam_max, max(lam))
            blocks.append((f, A, B, lam, V))
        tolW = _matlab_tol(E * N, lam_max)                                           # pinv's tolerance for the EN x EN matrix
        Nm = _zeros(u, u)
        rhs = _zeros(u)
        Ws, ws = [], []
        for i in range(N):
            f, A, B, lam, V = blocks[i]
            Wp = _zeros(E, E)
            for k in range(E):
                if lam[k] > tolW:
                    Wp = Wp + np.outer(V[:, k], V[:, k]) / lam[k]
            Wp = Wp + _EPS12 * np.eye(E, dtype=object)                               # :57
            w = -f - B.dot(xm[6 * i:6 * i + 6] - xi[6 * i:6 * i + 6])                 # :58
            WA = Wp.dot(A)
            Nm = Nm + A.T.dot(WA)
            rhs = rhs + WA.T.dot(w)
            Ws.append(Wp); ws.append(w)
        Mk = _zeros(u + c, u + c)
        Mk[:u, :u] = Nm
        Mk[:u, u:] = C.T
        Mk[u:, :u] = C
        b = np.concatenate([rhs, -g])
        Pm, kept = _pinv_sym(Mk + _EPS12 * np.eye(u + c, dtype=object))              # :67
        aux = Pm.dot(b)
        dt = aux[:u]
        v = _zeros(6 * N)
        for i in range(N):
            f, A, B, lam, V = blocks[i]
            v[6 * i:6 * i + 6] = -(B.T.dot(Ws[i].dot(A.dot(dt) - ws[i])))            # :69
        ndt = mp.sqrt(sum(d * d for d in dt))
        nres = mp.sqrt(sum(r * r for r in (xi - xm - v)))
        obj = sum(q * q for q in v)
        hist.append((float(ndt), float(obj), int(kept)))
        if ndt < tol and nres < tol:                                                 # :71-73 (dy is empty: norm 0)
            reason = 'converged'
            break
        if obj > objFunc:                                                            # :75, factor = 1
            reason = 'rose'
            break
        objFunc = obj
        xi = xm + v
        ti = ti + dt                                                                 # :80
    out = (to_float(ti), to_float(xi), it, reason)
    return out + (hist,) if return_history else out


def ressl_model(Ind):
    """Ressl's callback (ResslTFTPoseEstimation.m:110-177) as a model for gauss_helmert_mp: 20 parameters, 2 constraints"""
    def model(ti):
        S, e21, e31, mn, T, Ind2 = _ressl_unpack(ti, Ind)
        g = np.array([sum(e * e for e in e31) - 1, sum(s * s for s in S.reshape(9)) - 1], dtype=object)
        C = _zeros(2, 20)
        C[0, 17:20] = 2 * e31
        C[1, 0:9] = 2 * S.reshape(9, order='F')
        return T, _ressl_D(S, e21, e31, mn, Ind2), g, C
    return model


def gauss_helmert_ressl_mp(x, x_est, p0, Ind, it_max=400, return_history=False):
    return gauss_helmert_mp(x, x_est, p0, ressl_model(Ind), 20, 2, it_max, return_history)


# ---- Nordberg's callback (NordbergTFTPoseEstimation.m:128-222): 19 parameters (three axis-angle rotations, ten entries of the sparse
# ---- tensor), one constraint
def _cross_mp(v):
    z = mp.mpf(0)
    return np.array([[z, -v[2], v[1]], [v[2], z, -v[0]], [-v[1], v[0], z]], dtype=object)


def _transf_t_mp(T0, U, V, W):
    """NordbergTFTPoseEstimation.m:217-222"""
    T = _zeros(3, 3, 3)
    for i in range(3):
        T[:, :, i] = V.T.dot(U[0, i] * T0[:, :, 0] + U[1, i] * T0[:, :, 1] + U[2, i] * T0[:, :, 2]).dot(W)
    return T


def nordberg_model(ti):
    one = mp.mpf(1)
    I3 = np.array([[one if r == c else mp.mpf(0) for c in range(3)] for r in range(3)], dtype=object)
    o, vec, Rm = [], [], []
    for k in range(3):
        xk = ti[3 * k:3 * k + 3]
        ok = mp.sqrt(sum(e * e for e in xk))
        vk = xk / ok
        cm = _cross_mp(vk)
        o.append(ok); vec.append(vk)
        Rm.append(I3 + mp.sin(ok) * cm + (1 - mp.cos(ok)) * cm.dot(cm))              # :133-141 (Rodrigues)
    U, V, W = Rm
    paramT = ti[9:19]
    tsv = _zeros(27)
    tsv[O._NORD_IND] = paramT
    Ts = tsv.reshape(3, 3, 3, order='F')
    T = _transf_t_mp(Ts, U.T, V.T, W.T)
    J = _zeros(27, 19)
    for i in range(10):
        e = _zeros(27)
        e[O._NORD_IND[i]] = one
        J[:, i + 9] = _transf_t_mp(e.reshape(3, 3, 3, order='F'), U.T, V.T, W.T).reshape(27, order='F')
    dR = [[None] * 3 for _ in range(3)]
    for k in range(3):
        ok, vk = o[k], vec[k]
        cm = _cross_mp(vk)
        for i in range(3):
            ei = I3[:, i]
            dR[k][i] = (-vk[i] * mp.sin(ok) * I3 + vk[i] * mp.cos(ok) * cm
                        + mp.sin(ok) * (1 / ok) * (_cross_mp(ei) - vk[i] * cm)
                        + vk[i] * mp.sin(ok) * np.outer(vk, vk)
                        + (1 - mp.cos(ok)) * (1 / ok) * (np.outer(vk, ei) + np.outer(ei, vk) - 2 * vk[i] * np.outer(vk, vk)))   # :176-190
    for i in range(3):
        J[:, i] = _transf_t_mp(Ts, dR[0][i].T, V.T, W.T).reshape(27, order='F')
        J[:, i + 3] = _transf_t_mp(Ts, U.T, dR[1][i].T, W.T).reshape(27, order='F')
        J[:, i + 6] = _transf_t_mp(Ts, U.T, V.T, dR[2][i].T).reshape(27, order='F')
    g = np.array([sum(q * q for q in paramT) - 1], dtype=object)
    C = _zeros(1, 19)
    C[0, 9:19] = 2 * paramT
    return T, J, g, C


def nordberg_start(Corresp, CalM):
    """fp64 start of NordbergTFTPoseEstimation.m:47-96 (from the numpy oracle)"""
    x1, N1 = O.Normalize2Ddata(Corresp[0:2, :])
    x2, N2 = O.Normalize2Ddata(Corresp[2:4, :])
    x3, N3 = O.Normalize2Ddata(Corresp[4:6, :])
    T, P1, P2, P3 = O.linearTFT(x1, x2, x3)
    P1, P2, P3, param0 = O.nordberg_param0(T, P1, P2, P3)
    x, x_est = O._gh_initial_obs(P1, P2, P3, x1, x2, x3)
    return x, x_est, param0, (N1, N2, N3)


def nordberg_tail(param, normals, CalM, Corresp):
    """NordbergTFTPoseEstimation.m:104-124 in fp64"""
    Rm = []
    for k in range(3):
        ok = np.linalg.norm(param[3 * k:3 * k + 3])
        Rm.append(O._rodrigues(ok, param[3 * k:3 * k + 3] / ok))
    U, V, W = Rm
    tsv = np.zeros(27)
    tsv[O._NORD_IND] = param[9:19]
    T = O._transf_t(O._unvecT(tsv), U.T, V.T, W.T)
    T = O.transform_TFT(T, normals[0], normals[1], normals[2], 1)
    R_t_2, R_t_3 = O.R_t_from_TFT(T, CalM, Corresp)
    return R_t_2, R_t_3, T


def NordbergTFTPoseEstimation_mp(Corresp, CalM):
    x, x_est, p0, normals = nordberg_start(Corresp, CalM)
    p_opt, _, it, reason = gauss_helmert_mp(x, x_est, p0, nordberg_model, 19, 1)
    R2, R3, T = nordberg_tail(p_opt, normals, CalM, Corresp)
    return R2, R3, T, it, reason


def ressl_start(Corresp, CalM):
    """fp64 start of ResslTFTPoseEstimation.m:47-81 (from the numpy oracle): x, x_est, p0, Ind and the normalisations."""
    x1, N1 = O.Normalize2Ddata(Corresp[0:2, :])
    x2, N2 = O.Normalize2Ddata(Corresp[2:4, :])
    x3, N3 = O.Normalize2Ddata(Corresp[4:6, :])
    T, P1, P2, P3 = O.linearTFT(x1, x2, x3)
    e21 = P2[:, 3].copy()
    Ind = int(np.argmax(np.abs(e21)))
    e21 = e21 / e21[Ind]
    e31 = P3[:, 3].copy()
    e31 = e31 / np.linalg.norm(e31)
    S = np.stack([T[Ind, :, 0], T[Ind, :, 1], T[Ind, :, 2]], axis=1)
    aux = np.linalg.norm(S.reshape(9))
    S = S / aux
    T = T / aux
    Ind2 = [k for k in range(3) if k != Ind]
    mn = np.stack([e31 @ (T[:, :, i].T - np.outer(S[:, i], e21)) for i in range(3)], axis=0)[:, Ind2]
    x, x_est = O._gh_initial_obs(P1, P2, P3, x1, x2, x3)
    p = np.concatenate([S.reshape(9, order='F'), e21[Ind2], mn.reshape(6, order='F'), e31])
    return x, x_est, p, Ind, (N1, N2, N3)


def ressl_tail(p_opt, Ind, normals, CalM, Corresp):
    """ResslTFTPoseEstimation.m:87-103 in fp64: T from p_opt, de-normalisation, R_t_from_TFT."""
    Ind2 = [k for k in range(3) if k != Ind]
    S = p_opt[0:9].reshape(3, 3, order='F')
    e21 = np.ones(3); e21[Ind2] = p_opt[9:11]
    mn = np.zeros((3, 3)); mn[:, Ind2] = p_opt[11:17].reshape(3, 2, order='F')
    T = O._ressl_T(S, e21, p_opt[17:20], mn)
    T = O.transform_TFT(T, normals[0], normals[1], normals[2], 1)
    R_t_2, R_t_3 = O.R_t_from_TFT(T, CalM, Corresp)
    return R_t_2, R_t_3, T


def ResslTFTPoseEstimation_mp(Corresp, CalM):
    """R_t_2, R_t_3, T, iter, reason with the Gauss-Helmert loop in extended precision."""
    x, x_est, p0, Ind, normals = ressl_start(Corresp, CalM)
    p_opt, _, it, reason = gauss_helmert_ressl_mp(x, x_est, p0, Ind)
    R2, R3, T = ressl_tail(p_opt, Ind, normals, CalM, Corresp)
    return R2, R3, T, it, reason


# ---- Faugeras-Papadopoulo's callback (FaugPapaTFTPoseEstimation.m:87-159): all 27 tensor entries, 12 algebraic constraints
def _det3_mp(A):
    return (A[0, 0] * (A[1, 1] * A[2, 2] - A[1, 2] * A[2, 1]) - A[0, 1] * (A[1, 0] * A[2, 2] - A[1, 2] * A[2, 0])
            + A[0, 2] * (A[1, 0] * A[2, 1] - A[1, 1] * A[2, 0]))


def _minor_mp(A, i, j):
    """:156-159 (signed cofactor of a 3 x 3 matrix; i, j 0-based)"""
    r = [k for k in range(3) if k != i]
    c = [k for k in range(3) if k != j]
    d = A[r[0], c[0]] * A[r[1], c[1]] - A[r[0], c[1]] * A[r[1], c[0]]
    return d if (i + j) % 2 == 0 else -d


def _fp_stack_mp(T, idx3):
    out = _zeros(3, 3)
    for c, (a, b) in enumerate(idx3):
        out[c, :] = T[a, b, :]
    return out


def faugpapa_model(ti):
    T = ti.reshape(3, 3, 3, order='F')
    one = mp.mpf(1)
    D = _zeros(27, 27)
    for k in range(27):
        D[k, k] = one
    g = _zeros(12)
    C = _zeros(12, 27)
    for i in range(3):                                                               # :116-124: det(T_i) = 0
        g[i] = _det3_mp(T[:, :, i])
        for j in range(3):
            for k in range(3):
                C[i, j + 3 * k + 9 * i] = _minor_mp(T[:, :, i], j, k)
    i = -1
    for k2 in range(2):                                                              # :126-152: the nine extended-rank constraints
        for k3 in range(2):
            for l2 in range(k2 + 1, 3):
                for l3 in range(k3 + 1, 3):
                    i += 1
                    A1 = _fp_stack_mp(T, [(k2, k3), (k2, l3), (l2, l3)])
                    A2 = _fp_stack_mp(T, [(k2, k3), (l2, k3), (l2, l3)])
                    A3 = _fp_stack_mp(T, [(l2, k3), (k2, l3), (l2, l3)])
                    A4 = _fp_stack_mp(T, [(k2, k3), (l2, k3), (k2, l3)])
                    d1, d2, d3, d4 = _det3_mp(A1), _det3_mp(A2), _det3_mp(A3), _det3_mp(A4)
                    g[3 + i] = d1 * d2 - d3 * d4
                    for i1 in range(3):
                        C[3 + i, k2 + 3 * k3 + 9 * i1] = _minor_mp(A1, i1, 0) * d2 + d1 * _minor_mp(A2, i1, 0) - d3 * _minor_mp(A4, i1, 0)
                        C[3 + i, k2 + 3 * l3 + 9 * i1] = _minor_mp(A1, i1, 1) * d2 - _minor_mp(A3, i1, 1) * d4 - d3 * _minor_mp(A4, i1, 2)
                        C[3 + i, l2 + 3 * l3 + 9 * i1] = _minor_mp(A1, i1, 2) * d2 + d1 * _minor_mp(A2, i1, 2) - _minor_mp(A3, i1, 2) * d4
                        C[3 + i, l2 + 3 * k3 + 9 * i1] = d1 * _minor_mp(A2, i1, 1) - _minor_mp(A3, i1, 0) * d4 - d3 * _minor_mp(A4, i1, 1)
    return T, D, g, C


def faugpapa_start(Corresp, CalM):
    x1, N1 = O.Normalize2Ddata(Corresp[0:2, :])
    x2, N2 = O.Normalize2Ddata(Corresp[2:4, :])
    x3, N3 = O.Normalize2Ddata(Corresp[4:6, :])
    T, P1, P2, P3 = O.linearTFT(x1, x2, x3)
    x, x_est = O._gh_initial_obs(P1, P2, P3, x1, x2, x3)
    return x, x_est, O._vecT(T), (N1, N2, N3)


def FaugPapaTFTPoseEstimation_mp(Corresp, CalM):
    x, x_est, p0, normals = faugpapa_start(Corresp, CalM)
    p_opt, _, it, reason = gauss_helmert_mp(x, x_est, p0, faugpapa_model, 27, 12)
    T = O.transform_TFT(O._unvecT(p_opt), normals[0], normals[1], normals[2], 1)
    R_t_2, R_t_3 = O.R_t_from_TFT(T, CalM, Corresp)
    return R_t_2, R_t_3, T, it, reason


# ---- Ponce-Hebert Pi-matrix callback (PiPoseEstimation.m:109-182): 27 parameters (nine 3-vectors), 9 constraints; per correspondence
# ---- three epipolar equations and one trilinearity
def pi_model(ti):
    pi = ti
    pi21, pi31, pi41 = pi[0:3], pi[3:6], pi[6:9]
    pi12, pi32, pi42 = pi[9:12], pi[12:15], pi[15:18]
    pi13, pi23, pi43 = pi[18:21], pi[21:24], pi[24:27]
    F12 = np.outer(pi41, pi32) - np.outer(pi31, pi42)                                # :118-120
    F13 = np.outer(pi41, pi23) - np.outer(pi21, pi43)
    F23 = np.outer(pi42, pi13) - np.outer(pi12, pi43)
    dot = lambda a, b: sum(x * y for x, y in zip(a, b))
    g = np.array([dot(pi41, pi41) - 1, dot(pi42, pi42) - 1, dot(pi43, pi43) - 1,
                  dot(pi21, pi21) - 1, dot(pi32, pi32) - 1, dot(pi13, pi13) - 1,
                  dot(pi21, pi41), dot(pi32, pi42), dot(pi13, pi43)], dtype=object)   # :123-125
    C = _zeros(9, 27)                                                                # :128-137
    C[0, 6:9] = 2 * pi41
    C[1, 15:18] = 2 * pi42
    C[2, 24:27] = 2 * pi43
    C[3, 0:3] = 2 * pi21
    C[4, 12:15] = 2 * pi32
    C[5, 18:21] = 2 * pi13
    C[6, 0:3] = pi41; C[6, 6:9] = pi21
    C[7, 12:15] = pi42; C[7, 15:18] = pi32
    C[8, 18:21] = pi43; C[8, 24:27] = pi13

    def point_fn(o):
        one = mp.mpf(1)
        p1 = np.array([o[0], o[1], one], dtype=object)
        p2 = np.array([o[2], o[3], one], dtype=object)
        p3 = np.array([o[4], o[5], one], dtype=object)
        a21, a31, a41 = dot(pi21, p1), dot(pi31, p1), dot(pi41, p1)
        a12, a32, a42 = dot(pi12, p2), dot(pi32, p2), dot(pi42, p2)
        a13, a23, a43 = dot(pi13, p3), dot(pi23, p3), dot(pi43, p3)
        f = np.array([p1.dot(F12).dot(p2), p1.dot(F13).dot(p3), p2.dot(F23).dot(p3), a21 * a32 * a13 - a31 * a12 * a23], dtype=object)   # :152-153
        A = _zeros(4, 27)                                                            # :156-164
        A[0, 3:6] = -a42 * p1; A[0, 6:9] = a32 * p1
        A[0, 12:15] = a41 * p2; A[0, 15:18] = -a31 * p2
        A[1, 0:3] = -a43 * p1; A[1, 6:9] = a23 * p1
        A[1, 21:24] = a41 * p3; A[1, 24:27] = -a21 * p3
        A[2, 9:12] = -a43 * p2; A[2, 15:18] = a13 * p2
        A[2, 18:21] = a42 * p3; A[2, 24:27] = -a12 * p3
        A[3, 0:3] = p1 * (a32 * a13); A[3, 3:6] = -p1 * (a12 * a23)
        A[3, 9:12] = -(a31 * a23) * p2; A[3, 12:15] = (a21 * a13) * p2
        A[3, 18:21] = (a21 * a32) * p3; A[3, 21:24] = -(a31 * a12) * p3
        B = _zeros(4, 6)                                                             # :166-171
        B[0, 0:2] = F12.dot(p2)[0:2]; B[0, 2:4] = p1.dot(F12)[0:2]
        B[1, 0:2] = F13.dot(p3)[0:2]; B[1, 4:6] = p1.dot(F13)[0:2]
        B[2, 2:4] = F23.dot(p3)[0:2]; B[2, 4:6] = p2.dot(F23)[0:2]
        B[3, 0:2] = (pi21 * (a32 * a13) - pi31 * (a12 * a23))[0:2]
        B[3, 2:4] = (pi32 * (a21 * a13) - pi12 * (a31 * a23))[0:2]
        B[3, 4:6] = (pi13 * (a21 * a32) - pi23 * (a31 * a12))[0:2]
        return f, A, B
    return point_fn, g, C


def pi_start(Corresp, CalM):
    """fp64 start of PiPoseEstimation.m:50-88 (from the numpy oracle): x, x_est, pi0 and the normalisations"""
    x1, N1 = O.Normalize2Ddata(Corresp[0:2, :])
    x2, N2 = O.Normalize2Ddata(Corresp[2:4, :])
    x3, N3 = O.Normalize2Ddata(Corresp[4:6, :])
    pi0, x_est = O.PiPoseEstimation(Corresp, CalM, init_only=True)
    N = x1.shape[1]
    x = np.vstack([x1[0:2, :], x2[0:2, :], x3[0:2, :]]).reshape(6 * N, order='F')
    return x, x_est, pi0, (N1, N2, N3)


def PiPoseEstimation_mp(Corresp, CalM):
    x, x_est, p0, normals = pi_start(Corresp, CalM)
    p_opt, _, it, reason = gauss_helmert_mp(x, x_est, p0, pi_model, 27, 9)
    inv = np.linalg.inv
    Pi1 = p_opt[0:9].reshape(3, 3); Pi2 = p_opt[9:18].reshape(3, 3); Pi3 = p_opt[18:27].reshape(3, 3)   # :94-96
    P1 = np.zeros((3, 4)); P2 = np.zeros((3, 4)); P3 = np.zeros((3, 4))
    P1[:, 1:4] = inv(Pi1)
    P2[:, [0, 2, 3]] = inv(Pi2)
    P3[:, [0, 1, 3]] = inv(Pi3)
    R_t_2, R_t_3, _, T = O._pi_finish(P1, P2, P3, normals[0], normals[1], normals[2], CalM, Corresp)
    return R_t_2, R_t_3, T, it, reason


# ---- Ponce-Hebert Pi-matrix callback for collinear centres (PiColPoseEstimation.m:144-216): 27 parameters, 11 constraints; per
# ---- correspondence three epipolar equations and two trilinearities (5 x 5 weight blocks).  The sign of A(ind2+4,1:3) (:186), which is
# ---- not the derivative of f(ind2+4), is kept.
def picol_model(ti):
    pi = ti
    pi21, pi31, pi41 = pi[0:3], pi[3:6], pi[6:9]
    pi12, pi32, pi42 = pi[9:12], pi[12:15], pi[15:18]
    w3, pi33, pi43 = pi[18:21], pi[21:24], pi[24:27]
    F12 = np.outer(pi41, pi32) - np.outer(pi31, pi42)                                # :157-159
    F13 = np.outer(pi41, pi33) - np.outer(pi31, pi43)
    F23 = np.outer(pi42, pi33) - np.outer(pi32, pi43)
    dot = lambda a, b: sum(x * y for x, y in zip(a, b))
    g = np.array([dot(pi21, pi21) - 1, dot(pi12, pi12) - 1, dot(w3, w3) - 1, dot(pi33, pi33) - 1, dot(pi43, pi43) - 1,
                  dot(pi21, pi31), dot(pi21, pi41), dot(pi31, pi41), dot(pi12, pi32), dot(pi12, pi42), dot(pi32, pi42)], dtype=object)   # :162-165
    C = _zeros(11, 27)                                                               # :168-173
    C[0, 0:3] = 2 * pi21; C[1, 9:12] = 2 * pi12
    C[2, 18:21] = 2 * w3; C[3, 21:24] = 2 * pi33; C[4, 24:27] = 2 * pi43
    C[5, 0:3] = pi31; C[5, 3:6] = pi21
    C[8, 9:12] = pi32; C[8, 12:15] = pi12
    C[6, 0:3] = pi41; C[6, 6:9] = pi21
    C[9, 9:12] = pi42; C[9, 15:18] = pi12
    C[7, 3:6] = pi41; C[7, 6:9] = pi31
    C[10, 12:15] = pi42; C[10, 15:18] = pi32

    def point_fn(o):
        one = mp.mpf(1)
        p1 = np.array([o[0], o[1], one], dtype=object)
        p2 = np.array([o[2], o[3], one], dtype=object)
        p3 = np.array([o[4], o[5], one], dtype=object)
        a21, a31, a41 = dot(pi21, p1), dot(pi31, p1), dot(pi41, p1)
        a12, a32, a42 = dot(pi12, p2), dot(pi32, p2), dot(pi42, p2)
        aw3, a33, a43 = dot(w3, p3), dot(pi33, p3), dot(pi43, p3)
        f = np.array([p1.dot(F12).dot(p2), p1.dot(F13).dot(p3), p2.dot(F23).dot(p3),
                      a31 * a32 * aw3 + (a31 * a12 - a21 * a32) * a33,
                      a41 * a42 * aw3 + (a41 * a12 - a21 * a42) * a43], dtype=object)   # :186-188
        A = _zeros(5, 27)                                                            # :191-206
        A[0, 3:6] = -a42 * p1; A[0, 6:9] = a32 * p1; A[0, 12:15] = a41 * p2; A[0, 15:18] = -a31 * p2
        A[1, 3:6] = -a43 * p1; A[1, 6:9] = a33 * p1; A[1, 21:24] = a41 * p3; A[1, 24:27] = -a31 * p3
        A[2, 12:15] = -a43 * p2; A[2, 15:18] = a33 * p2; A[2, 21:24] = a42 * p3; A[2, 24:27] = -a32 * p3
        A[3, 0:3] = p1 * (a32 * a33); A[3, 3:6] = p1 * (a32 * aw3 + a12 * a33)
        A[3, 9:12] = p2 * (a31 * a33); A[3, 12:15] = p2 * (a31 * aw3 - a21 * a33)
        A[3, 18:21] = p3 * (a31 * a32); A[3, 21:24] = p3 * (a31 * a12 - a21 * a32)
        A[4, 0:3] = -p1 * (a42 * a43); A[4, 6:9] = p1 * (a42 * aw3 + a12 * a43)
        A[4, 9:12] = p2 * (a41 * a43); A[4, 15:18] = p2 * (a41 * aw3 - a21 * a43)
        A[4, 18:21] = p3 * (a41 * a42); A[4, 24:27] = p3 * (a41 * a12 - a21 * a42)
        B = _zeros(5, 6)                                                             # :208-218
        B[0, 0:2] = F12.dot(p2)[0:2]; B[0, 2:4] = p1.dot(F12)[0:2]
        B[1, 0:2] = F13.dot(p3)[0:2]; B[1, 4:6] = p1.dot(F13)[0:2]
        B[2, 2:4] = F23.dot(p3)[0:2]; B[2, 4:6] = p2.dot(F23)[0:2]
        B[3, 0:2] = (pi31 * (a32 * aw3 + a12 * a33) - pi21 * (a32 * a33))[0:2]
        B[3, 2:4] = (pi32 * (a31 * aw3) + (pi12 * a31 - pi32 * a21) * a33)[0:2]
        B[3, 4:6] = (w3 * (a31 * a32) + pi33 * (a31 * a12 - a21 * a32))[0:2]
        B[4, 0:2] = (pi41 * (a42 * aw3) + (pi41 * a12 - pi21 * a42) * a43)[0:2]
        B[4, 2:4] = (pi42 * (a41 * aw3) + (pi12 * a41 - pi42 * a21) * a43)[0:2]
        B[4, 4:6] = (w3 * (a41 * a42) + pi43 * (a41 * a12 - a21 * a42))[0:2]
        return f, A, B
    return point_fn, g, C


def picol_start(Corresp, CalM, null=None, cam_signs=(1.0, 1.0)):
    """fp64 start of PiColPoseEstimation.m:50-115 (from the numpy oracle) under a stated convention for the null vectors / camera signs"""
    x1, N1 = O.Normalize2Ddata(Corresp[0:2, :])
    x2, N2 = O.Normalize2Ddata(Corresp[2:4, :])
    x3, N3 = O.Normalize2Ddata(Corresp[4:6, :])
    kw = dict(cam_signs=cam_signs, init_only=True)
    if null is not None:
        kw["null"] = null
    pi0, x_est = O.PiColPoseEstimation(Corresp, CalM, **kw)
    N = x1.shape[1]
    x = np.vstack([x1[0:2, :], x2[0:2, :], x3[0:2, :]]).reshape(6 * N, order='F')
    return x, x_est, pi0, (N1, N2, N3)


def PiColPoseEstimation_mp(Corresp, CalM, null=None, cam_signs=(1.0, 1.0)):
    x, x_est, p0, normals = picol_start(Corresp, CalM, null, cam_signs)
    p_opt, _, it, reason = gauss_helmert_mp(x, x_est, p0, picol_model, 27, 11)
    inv = np.linalg.inv
    Pi1 = p_opt[0:9].reshape(3, 3); Pi2 = p_opt[9:18].reshape(3, 3); Pi3 = p_opt[18:27].reshape(3, 3)   # :121-123
    P1 = np.zeros((3, 4)); P2 = np.zeros((3, 4)); P3 = np.zeros((3, 4))
    P1[:, 1:4] = inv(Pi1)
    P2[:, [0, 2, 3]] = inv(Pi2)
    P3[:, 1:4] = inv(Pi3); P3[:, 0] = -P3[:, 1]                                      # :127
    R_t_2, R_t_3, _, T = O._pi_finish(P1, P2, P3, normals[0], normals[1], normals[2], CalM, Corresp)
    return R_t_2, R_t_3, T, it, reason

"""
TEST INFRASTRUCTURE ONLY -- numpy/LAPACK restatement of the TFT_vs_Fund hot path.

This file is the *checker*: a literal, function-by-function restatement of the
reference's MATLAB algorithm in numpy (LAPACK-backed `svd`, `solve`), written
so that every line can be audited against the `.m` file it follows.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it.  The product path (`tft_vs_fund_amd/`, the HIP library) never
does, and never falls back to it.

PARITY UNPINNED: the reference is MATLAB-only, ships no tests, golden vectors
or fixtures for this path, and cannot be executed in the build container
(no matlab / octave / mex).  The closed-source built-ins `svd`, `pinv`,
`rank`, `null`, `inv`, `det`, `mldivide`, `mpower` (MATLAB release unpinned,
README.txt:9,33) are restated here from their documented semantics:
  * svd   -> LAPACK via numpy; descending singular values, sign of singular
             vectors unspecified (all consumers are sign-invariant or fix the
             sign themselves);
  * pinv / rank / null tolerance -> max(size(A)) * eps(norm(A,2));
  * inv, det, mldivide (square) -> LU with partial pivoting (numpy);
  * A^(-1/2) for symmetric positive definite A -> eigen-decomposition.
What pins this oracle instead is listed in tests/test_oracle_pins.py:
known-answer properties that follow from the reference's own ground truth
(noise-free scenes recover R_t0, rank(E) = 15, cheirality votes = +-2N / 0,
EPFL 1-px inlier counts).

All file:line citations are relative to the reference tree.
Arrays follow MATLAB shapes: Corresp is 6xN, CalM is 9x3, T is 3x3x3 indexed
T[j,k,i] == T(j+1,k+1,i+1); vec(T) is column-major (`order='F'`).
"""
import numpy as np

_EPS12 = 1e-12


# --------------------------------------------------------------------------
# MATLAB built-ins restated
# --------------------------------------------------------------------------
def _svd(A):
    """[U,S,V] = svd(A) (full).  Returns U, s (vector), V (not V')."""
    U, s, Vh = np.linalg.svd(A, full_matrices=True)
    return U, s, Vh.T


def _svdV(A):
    """[~,~,V] = svd(A): only V is consumed, so the thin factorisation is
    enough whenever rows >= cols (V is then n x n: 4N x 27 with N >= 7, 3x3,
    2M x 4).  With fewer rows than columns (linearF at N = 8: 8 x 9) MATLAB's
    full svd still returns a 9 x 9 V whose last column spans the null space,
    so the full factorisation is taken."""
    _, _, Vh = np.linalg.svd(A, full_matrices=A.shape[0] < A.shape[1])
    assert Vh.shape[0] == A.shape[1]
    return Vh.T


def _tol(A, s):
    # MATLAB: max(size(A)) * eps(norm(A)); norm(A) = largest singular value
    return max(A.shape) * np.spacing(s[0] if s.size else 0.0)


def rank(A):
    s = np.linalg.svd(A, compute_uv=False)
    return int(np.sum(s > _tol(A, s)))


def pinv(A):
    U, s, Vh = np.linalg.svd(A, full_matrices=False)
    r = int(np.sum(s > _tol(A, s)))
    return (Vh[:r, :].T * (1.0 / s[:r])) @ U[:, :r].T


def null(A):
    U, s, V = _svd(A)
    r = int(np.sum(s > _tol(A, s)))
    return V[:, r:]


def _sign(x):
    return np.sign(x)  # sign(0) == 0 in both MATLAB and numpy


def _mpower_invsqrt(A):
    """A^(-1/2) for the symmetric positive definite Gram matrices of
    NordbergTFTPoseEstimation.m:68-70 (MATLAB mpower -> eig)."""
    w, Q = np.linalg.eigh((A + A.T) / 2)
    return (Q * (w ** -0.5)) @ Q.T


# --------------------------------------------------------------------------
# auxiliar_functions
# --------------------------------------------------------------------------
def crossM(v):
    """auxiliar_functions/crossM.m:22"""
    v = np.asarray(v).reshape(3)
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=float)


def Normalize2Ddata(points):
    """auxiliar_functions/Normalize2Ddata.m:33-39 (returns 2xN, quirk #1)."""
    n = points.shape[1]
    points0 = np.sum(points, axis=1, keepdims=True) / n
    norm0 = np.sum(np.sqrt(np.sum((points - points0) ** 2, axis=0))) / n
    N_matrix = np.diag([np.sqrt(2) / norm0, np.sqrt(2) / norm0, 1.0])
    N_matrix[0:2, 2] = -np.sqrt(2) * points0[:, 0] / norm0
    new_points = N_matrix[0:2, :] @ np.vstack([points, np.ones((1, n))])
    return new_points, N_matrix


def triangulation3D(Pcam, image_points):
    """auxiliar_functions/triangulation3D.m:32-64.  Pcam: list of M 3x4;
    image_points 2M x N or 3M x N.  Returns 4xN unit-norm homogeneous points
    (sign free, not dehomogenised)."""
    M = len(Pcam)
    if M < 2:
        return None
    N = image_points.shape[1]
    if image_points.shape[0] == 3 * M:
        aux = image_points.reshape(3, N * M, order='F')
        aux = aux[0:2, :] / aux[2:3, :]
        image_points = aux.reshape(2 * M, N, order='F')
    elif image_points.shape[0] != 2 * M:
        return None
    ls = np.zeros((N, 2 * M, 4))
    for i in range(M):
        x = image_points[2 * i, :]
        y = image_points[2 * i + 1, :]
        P = np.asarray(Pcam[i], dtype=float)
        # [0 -1 y; 1 0 -x] * P   (triangulation3D.m:58-59)
        ls[:, 2 * i, :] = 0 * P[0][None, :] + (-1) * P[1][None, :] + y[:, None] * P[2][None, :]
        ls[:, 2 * i + 1, :] = 1 * P[0][None, :] + 0 * P[1][None, :] + (-x)[:, None] * P[2][None, :]
    _, _, Vh = np.linalg.svd(ls, full_matrices=False)  # batched LAPACK, one SVD per point
    return Vh[:, 3, :].T.copy()


def project3Dpoints(Points3D, Pcam):
    """auxiliar_functions/project3Dpoints.m:28-35"""
    M = len(Pcam)
    N = Points3D.shape[1]
    Corresp = np.zeros((2 * M, N))
    for m in range(M):
        x = Pcam[m] @ np.vstack([Points3D, np.ones((1, N))])
        Corresp[2 * m:2 * m + 2, :] = x[0:2, :] / x[2:3, :]
    return Corresp


def ReprError(ProjM, Corresp, Points3D=None):
    """auxiliar_functions/ReprError.m:39-65"""
    N = Corresp.shape[1]
    M = len(ProjM)
    if Points3D is None:
        P3 = triangulation3D(ProjM, Corresp)
    elif Points3D.shape[0] == 3:
        P3 = np.vstack([Points3D, np.ones((1, N))])
    else:
        P3 = Points3D
    if Corresp.shape[0] == 3 * M:
        C = Corresp.reshape(3, N * M, order='F')
        C = C[0:2, :] / C[2:3, :]
    else:
        C = Corresp.reshape(2, N * M, order='F')
    P = np.vstack(ProjM)
    Ce = (P @ P3).reshape(3, M * N, order='F')
    Ce = Ce[0:2, :] / Ce[2:3, :]
    return float(np.sqrt(np.mean(np.sum((Ce - C) ** 2, axis=0))))


def AngError(R_t_true, R_t_est):
    """auxiliar_functions/AngError.m:21-28.  The acos argument is not clamped: when rounding pushes it above 1
    MATLAB's acos returns a purely imaginary number and abs() its magnitude (a tiny angle), restated here with the
    complex arccos."""
    R_true, t_true = R_t_true[:, 0:3], R_t_true[:, 3]
    R_est, t_est = R_t_est[:, 0:3], R_t_est[:, 3]
    rot_err = float(abs(180 * np.arccos(complex((np.trace(R_true.T @ R_est) - 1) / 2)) / np.pi))
    t_err = float(abs(180 * np.arccos(complex(np.dot(t_est / np.linalg.norm(t_est),
                                                       t_true / np.linalg.norm(t_true)))) / np.pi))
    return rot_err, t_err


# --------------------------------------------------------------------------
# TFT_methods
# --------------------------------------------------------------------------
def _vecT(T):
    return T.reshape(27, order='F')


def _unvecT(t):
    return np.asarray(t, dtype=float).reshape(3, 3, 3, order='F')


def transform_TFT(T_old, M1, M2, M3, inverse=0):
    """TFT_methods/transform_TFT.m:32-49"""
    T_new = np.zeros((3, 3, 3))
    if inverse == 0:
        M1i = np.linalg.inv(M1)
        for i in range(3):
            T_new[:, :, i] = M2 @ (M1i[0, i] * T_old[:, :, 0] + M1i[1, i] * T_old[:, :, 1]
                                   + M1i[2, i] * T_old[:, :, 2]) @ M3.T
    elif inverse == 1:
        M2i = np.linalg.inv(M2)
        M3i = np.linalg.inv(M3)
        for i in range(3):
            T_new[:, :, i] = M2i @ (M1[0, i] * T_old[:, :, 0] + M1[1, i] * T_old[:, :, 1]
                                    + M1[2, i] * T_old[:, :, 2]) @ M3i.T
    return T_new / np.linalg.norm(_vecT(T_new))


def TFT_from_P(P1, P2, P3):
    """TFT_methods/TFT_from_P.m:25-33"""
    T = np.zeros((3, 3, 3))
    for i in range(3):
        rows = [r for r in range(3) if r != i]
        for j in range(3):
            for k in range(3):
                Mx = np.vstack([P1[rows, :], P2[j:j + 1, :], P3[k:k + 1, :]])
                T[j, k, i] = (-1) ** (i + 2) * np.linalg.det(Mx)
    return T / np.linalg.norm(_vecT(T))


def _linearTFT_system(p1, p2, p3):
    """TFT_methods/linearTFT.m:36-62: the 4N x 27 DLT matrix."""
    N = p1.shape[1]
    if p1.shape[0] == 3:
        p1 = p1[0:2, :] / p1[2:3, :]
        p2 = p2[0:2, :] / p2[2:3, :]
        p3 = p3[0:2, :] / p3[2:3, :]
    x1, y1 = p1[0], p1[1]
    x2, y2 = p2[0], p2[1]
    x3, y3 = p3[0], p3[1]
    o = np.ones(N)
    z = np.zeros(N)
    A = np.zeros((4 * N, 27))
    A[0::4, :] = np.stack([x1, z, -x1 * x2, z, z, z, -x1 * x3, z, x1 * x2 * x3,
                           y1, z, -x2 * y1, z, z, z, -x3 * y1, z, x2 * x3 * y1,
                           o, z, -x2, z, z, z, -x3, z, x2 * x3], axis=1)
    A[1::4, :] = np.stack([z, x1, -x1 * y2, z, z, z, z, -x1 * x3, x1 * x3 * y2,
                           z, y1, -y1 * y2, z, z, z, z, -x3 * y1, x3 * y1 * y2,
                           z, o, -y2, z, z, z, z, -x3, x3 * y2], axis=1)
    A[2::4, :] = np.stack([z, z, z, x1, z, -x1 * x2, -x1 * y3, z, x1 * x2 * y3,
                           z, z, z, y1, z, -x2 * y1, -y1 * y3, z, x2 * y1 * y3,
                           z, z, z, o, z, -x2, -y3, z, x2 * y3], axis=1)
    A[3::4, :] = np.stack([z, z, z, z, x1, -x1 * y2, z, -x1 * y3, x1 * y2 * y3,
                           z, z, z, z, y1, -y1 * y2, z, -y1 * y3, y1 * y2 * y3,
                           z, z, z, z, o, -y2, z, -y3, y2 * y3], axis=1)
    return A


def _epipoles_from_T(T):
    """linearTFT.m:71-79 / R_t_from_TFT.m:47-55 without the sign fix:
    returns the full V of the two stacking SVDs' last columns."""
    v = [_svdV(T[:, :, i])[:, -1] for i in range(3)]
    V31 = _svdV(np.stack(v, axis=1).T)
    v = [_svdV(T[:, :, i].T)[:, -1] for i in range(3)]
    V21 = _svdV(np.stack(v, axis=1).T)
    return V21, V31


def linearTFT(p1, p2, p3, return_debug=False):
    """TFT_methods/linearTFT.m:33-91.  Returns T, P1, P2, P3."""
    A = _linearTFT_system(p1, p2, p3)
    V = _svdV(A)                                        # :64
    t = V[:, -1]
    T = _unvecT(t)                                      # :67
    V21, V31 = _epipoles_from_T(T)                      # :71-79
    epi31 = V31[:, -1]
    epi21 = V21[:, -1]
    # [~,~,V] = svd(...); epi = V(:,3): the SIGN of each epipole is whatever the SVD routine returns.  It flips the blocks
    # a(1:9) / a(10:18) below with it (P2 = [A | e21], P3 = [B | e31] stay a valid camera pair of the same T), which is a LINEAR
    # change of every later parameterisation but Nordberg's -- its three rotations are built from these cameras through
    # orth(), so its Gauss-Helmert iterates (not its fixed point) depend on the convention.  EPIPOLE_SIGNS replays the others.
    if EPIPOLE_SIGNS is not None:
        epi21 = epi21 * EPIPOLE_SIGNS[0]
        epi31 = epi31 * EPIPOLE_SIGNS[1]
    E = np.hstack([np.kron(np.eye(3), np.kron(epi31.reshape(3, 1), np.eye(3))),
                   -np.kron(np.eye(9), epi21.reshape(3, 1))])   # :82
    U, S, V = _svd(E)
    rk = rank(E)                                        # :83
    Up, Vp, Sp = U[:, :rk], V[:, :rk], np.diag(S[:rk])
    tp = _svdV(A @ Up)[:, -1]                           # :84
    if EPIPOLE_SIGNS is not None and len(EPIPOLE_SIGNS) > 2:
        tp = tp * EPIPOLE_SIGNS[2]                      # ... and so is the sign of this V(:,end): it flips T, a(1:9) and a(10:18) together
    t = Up @ tp                                         # :85
    a = Vp @ np.linalg.inv(Sp) @ tp                     # :86
    P1 = np.eye(3, 4)
    P2 = np.hstack([a[0:9].reshape(3, 3, order='F'), epi21.reshape(3, 1)])
    P3 = np.hstack([a[9:18].reshape(3, 3, order='F'), epi31.reshape(3, 1)])
    T = _unvecT(t)
    if return_debug:
        return T, P1, P2, P3, dict(rankE=rk, epi21=epi21, epi31=epi31,
                                   sv_A=np.linalg.svd(A, compute_uv=False))
    return T, P1, P2, P3


EPIPOLE_SIGNS = None    # test hook, see linearTFT


def set_epipole_signs(signs):
    """signs: None (numpy's LAPACK as is) or (sign of e21, sign of e31[, sign of the constrained solution tp]) applied inside linearTFT."""
    global EPIPOLE_SIGNS
    EPIPOLE_SIGNS = signs


E_SVD_SIGNS = None      # test hook, see _recover_R_t_core
_recover_call = 0


def set_E_svd_signs(signs):
    """signs: None, or a list of (sign U(:,3), sign V(:,3)) pairs, one per recover_R_t call of a method (two calls)."""
    global E_SVD_SIGNS, _recover_call
    E_SVD_SIGNS = signs
    _recover_call = 0


def _recover_R_t_core(E21, P1cam, K2, x1, x2, return_debug=False):
    """R_t_from_TFT.m:84-104 (== LinearFPoseEstimation.m:87-107 after E21 is
    formed).  Candidate order (R,t),(R,-t),(Rp,-t),(Rp,t); `>=` keeps the
    later candidate on ties; all-negative scores leave the result unassigned
    (None here; a MATLAB runtime error in the reference)."""
    W = np.array([[0., -1, 0], [1, 0, 0], [0, 0, 1]])
    U, _, V = _svd(E21)
    # [U,~,V] = svd(E21): the signs of U(:,3) and V(:,3) (E21 has rank 2) are whatever the SVD routine returns; they
    # permute the candidate order below, which decides TIES of the `>=` rule.  E_SVD_SIGNS lets a test replay the other
    # conventions (entry `call` of the list for the call-th recover_R_t of a method; None = numpy's LAPACK as is).
    global _recover_call
    if E_SVD_SIGNS is not None:
        su, sv = E_SVD_SIGNS[_recover_call % len(E_SVD_SIGNS)]
        U = U.copy(); V = V.copy()
        U[:, 2] *= su
        V[:, 2] *= sv
    _recover_call += 1
    R = U @ W @ V.T
    Rp = U @ W.T @ V.T
    R = R * _sign(np.linalg.det(R))
    Rp = Rp * _sign(np.linalg.det(Rp))
    t = U[:, 2].copy()
    num_points_seen = 0
    R_f = t_f = None
    votes = []
    for k in range(1, 5):
        if k == 2 or k == 4:
            t = -t
        elif k == 3:
            R = Rp
        Rt = np.hstack([R, t.reshape(3, 1)])
        X1 = triangulation3D([P1cam, K2 @ Rt], np.vstack([x1, x2]))
        X1 = X1 / X1[3:4, :]
        X2 = Rt @ X1
        score = np.sum(_sign(X1[2, :]) + _sign(X2[2, :]))
        votes.append(score)
        if score >= num_points_seen:
            R_f, t_f = R.copy(), t.copy()
            num_points_seen = score
    if return_debug:
        return R_f, t_f, votes
    return R_f, t_f


def _t3_scale(K1, K2, K3, R2, t2, R3, t3, Corresp):
    """R_t_from_TFT.m:68-74 == LinearFPoseEstimation.m:64-70"""
    N = Corresp.shape[1]
    u3 = K3 @ t3
    X = triangulation3D([K1 @ np.eye(3, 4), K2 @ np.hstack([R2, t2.reshape(3, 1)])], Corresp[0:4, :])
    X = X[0:3, :] / X[3:4, :]
    X3 = K3 @ R3 @ X
    p3 = np.vstack([Corresp[4:6, :], np.ones((1, N))])
    U3 = np.tile(u3.reshape(3, 1), (1, N))
    c1 = np.cross(p3, X3, axis=0)
    c2 = np.cross(p3, U3, axis=0)
    lam = -np.sum(np.sum(c1 * c2, axis=0)) / np.sum(np.sum(c2 ** 2))
    return lam


def R_t_from_TFT(T, CalM, Corresp, return_debug=False):
    """TFT_methods/R_t_from_TFT.m:40-76"""
    K1, K2, K3 = CalM[0:3, :], CalM[3:6, :], CalM[6:9, :]
    T = transform_TFT(T, K1, K2, K3, 1)                 # :44
    V21, V31 = _epipoles_from_T(T)
    epi31 = V31[:, -1] * _sign(V31[-1, -1])             # :50  V(end) == V(3,3)
    epi21 = V21[:, -1] * _sign(V21[-1, -1])             # :55
    E21 = crossM(epi21) @ np.stack([T[:, :, i] @ epi31 for i in range(3)], axis=1)       # :57
    E31 = -crossM(epi31) @ np.stack([T[:, :, i].T @ epi21 for i in range(3)], axis=1)    # :58
    P1cam = K1 @ np.eye(3, 4)
    R2, t2, v2 = _recover_R_t_core(E21, P1cam, K2, Corresp[0:2, :], Corresp[2:4, :], True)
    R3, t3, v3 = _recover_R_t_core(E31, P1cam, K3, Corresp[0:2, :], Corresp[4:6, :], True)
    if R2 is None or R3 is None:
        if return_debug:
            return None, None, dict(votes2=v2, votes3=v3)
        return None, None
    lam = _t3_scale(K1, K2, K3, R2, t2, R3, t3, Corresp)
    t3 = lam * t3
    R_t_2 = np.hstack([R2, t2.reshape(3, 1)])
    R_t_3 = np.hstack([R3, t3.reshape(3, 1)])
    if return_debug:
        return R_t_2, R_t_3, dict(votes2=v2, votes3=v3, lam=lam, E21=E21, E31=E31,
                                  epi21=epi21, epi31=epi31)
    return R_t_2, R_t_3


def _final_reconst(CalM, R_t_2, R_t_3, Corresp):
    """LinearTFTPoseEstimation.m:59-60 and the identical tails of the other wrappers."""
    Rec = triangulation3D([CalM[0:3, :] @ np.eye(3, 4), CalM[3:6, :] @ R_t_2, CalM[6:9, :] @ R_t_3], Corresp)
    return Rec[0:3, :] / Rec[3:4, :]


def LinearTFTPoseEstimation(Corresp, CalM):
    """TFT_methods/LinearTFTPoseEstimation.m:44-62"""
    x1, Normal1 = Normalize2Ddata(Corresp[0:2, :])
    x2, Normal2 = Normalize2Ddata(Corresp[2:4, :])
    x3, Normal3 = Normalize2Ddata(Corresp[4:6, :])
    T, _, _, _ = linearTFT(x1, x2, x3)
    T = transform_TFT(T, Normal1, Normal2, Normal3, 1)
    R_t_2, R_t_3 = R_t_from_TFT(T, CalM, Corresp)
    Reconst = _final_reconst(CalM, R_t_2, R_t_3, Corresp)
    return R_t_2, R_t_3, Reconst, T, 0


# --------------------------------------------------------------------------
# F_methods
# --------------------------------------------------------------------------
def linearF(p1, p2):
    """F_methods/linearF.m:32-62"""
    N = p1.shape[1]
    if N != p2.shape[1] or N < 8:
        raise ValueError('At least 8 correspondences are necessary to compute the fundamental matrix linearly')
    if p1.shape[0] == 3:
        p1 = p1[0:2, :] / p1[2:3, :]
        p2 = p2[0:2, :] / p2[2:3, :]
    p1, Normal1 = Normalize2Ddata(p1[0:2, :])
    p2, Normal2 = Normalize2Ddata(p2[0:2, :])
    A = np.stack([p1[0] * p2[0], p1[0] * p2[1], p1[0], p1[1] * p2[0],
                  p1[1] * p2[1], p1[1], p2[0], p2[1], np.ones(N)], axis=1)
    V = _svdV(A)
    F = V[:, -1].reshape(3, 3, order='F')
    F = Normal2.T @ F @ Normal1
    U, D, V = _svd(F)
    D = D.copy()
    D[2] = 0
    return U @ np.diag(D) @ V.T


def LinearFPoseEstimation(Corresp, CalM):
    """F_methods/LinearFPoseEstimation.m:42-78"""
    K1, K2, K3 = CalM[0:3, :], CalM[3:6, :], CalM[6:9, :]
    x1, Normal1 = Normalize2Ddata(Corresp[0:2, :])
    x2, Normal2 = Normalize2Ddata(Corresp[2:4, :])
    x3, Normal3 = Normalize2Ddata(Corresp[4:6, :])
    F21 = linearF(x1, x2)
    F31 = linearF(x1, x3)
    F21 = Normal2.T @ F21 @ Normal1
    F31 = Normal3.T @ F31 @ Normal1
    P1cam = np.hstack([K1, np.zeros((3, 1))])
    R2, t2 = _recover_R_t_core(K2.T @ F21 @ K1, P1cam, K2, Corresp[0:2, :], Corresp[2:4, :])
    R3, t3 = _recover_R_t_core(K3.T @ F31 @ K1, P1cam, K3, Corresp[0:2, :], Corresp[4:6, :])
    lam = _t3_scale(K1, K2, K3, R2, t2, R3, t3, Corresp)
    t3 = lam * t3
    R_t_2 = np.hstack([R2, t2.reshape(3, 1)])
    R_t_3 = np.hstack([R3, t3.reshape(3, 1)])
    Reconst = _final_reconst(CalM, R_t_2, R_t_3, Corresp)
    T = TFT_from_P(K1 @ np.eye(3, 4), K2 @ R_t_2, K3 @ R_t_3)
    return R_t_2, R_t_3, Reconst, T, 0


# --------------------------------------------------------------------------
# Optimization/Gauss_Helmert.m
# --------------------------------------------------------------------------
def Gauss_Helmert(func, x0, t0, y0, x, P=None, return_debug=False):
    """Optimization/Gauss_Helmert.m:38-83.  In every caller restated here
    P = eye(6N) and y is empty; pinv(eye) and inv(eye) are exactly eye (the
    SVD of an identity is exact), so those products are elided when P is None.
    """
    it_max = 400
    tol = 1e-6
    xi, yi, ti = x0.copy(), y0.copy(), t0.copy()
    u = t0.shape[0]
    s = y0.shape[0]
    v0 = x0 - x
    objFunc = float(v0 @ v0) if P is None else float(v0 @ P @ v0)
    factor = 1
    reason = 'itmax'
    it = 0
    for it in range(1, it_max + 1):
        f, g, A, B, C, D = func(xi, ti, yi)
        c2 = C.shape[0]
        W = B @ B.T if P is None else B @ pinv(P) @ B.T                     # :52
        if not np.all(np.isfinite(W)):
            reason = 'nanW'
            break
        W = pinv(W + _EPS12 * np.eye(W.shape[0]))                            # :57
        W = W + _EPS12 * np.eye(W.shape[0])
        w = -f - B @ (x - xi)                                                # :58
        M = np.block([[A.T @ W @ A, np.zeros((u, s)), C.T],
                      [np.zeros((s, u + s)), D.T],
                      [C, D, np.zeros((c2, c2))]])                           # :59-61
        b = np.concatenate([A.T @ W @ w, np.zeros(s), -g])                   # :62
        if not np.all(np.isfinite(M)):
            reason = 'nanM'
            break
        aux = pinv(M + _EPS12 * np.eye(M.shape[0])) @ b                      # :67
        dt = aux[0:u]
        dy = aux[u:u + s]
        v = -(B.T @ (W @ (A @ dt - w)))                                      # :69
        if P is not None:
            v = np.linalg.inv(P) @ v
        if np.linalg.norm(dt) < tol and np.linalg.norm(dy) < tol and np.linalg.norm(xi - x - v) < tol:
            reason = 'converged'
            break                                                            # :71-73
        obj = float(v @ v) if P is None else float(v @ P @ v)
        if obj > objFunc * factor:                                           # :75
            reason = 'rose'
            break
        objFunc = obj
        xi = x + v
        ti = ti + dt
        yi = yi + dy                                                         # :80
    if return_debug:
        return xi, ti, yi, it, reason
    return xi, ti, yi, it


def _trilinear_blocks(x, T):
    """The per-point block shared by ResslTFTPoseEstimation.m:141-161,
    NordbergTFTPoseEstimation.m:154-170, FaugPapaTFTPoseEstimation.m:96-112.
    x is the 6N observation vector, T the current tensor."""
    N = x.shape[0] // 6
    f = np.zeros(4 * N)
    Ap = np.zeros((4 * N, 27))
    B = np.zeros((4 * N, 6 * N))
    T1, T2, T3 = T[:, :, 0], T[:, :, 1], T[:, :, 2]
    J3 = T[2, :, :].T.copy()      # rows T_i(3,:)   == reshape(T(3,:,:),3,3).'   (Ressl :123)
    K3 = T[:, 2, :].copy()        # cols T_i(:,3)   == reshape(T(:,3,:),3,3)     (Ressl :124)
    sw = np.array([[0., 1], [1, 0]])
    for i in range(N):
        ind = 6 * i
        x1 = x[ind:ind + 2]
        x2 = x[ind + 2:ind + 4]
        x3 = x[ind + 4:ind + 6]
        h1 = np.array([x1[0], x1[1], 1.0])
        ind2 = 4 * i
        S2 = np.array([[0, -1.], [-1, 0], [x2[1], x2[0]]])
        S3 = np.array([[0, -1.], [-1, 0], [x3[1], x3[0]]])
        f[ind2:ind2 + 4] = (S2.T @ (x1[0] * T1 + x1[1] * T2 + T3) @ S3).reshape(4, order='F')
        Ap[ind2:ind2 + 4, :] = np.kron(S3, S2).T @ np.kron(h1.reshape(1, 3), np.eye(9))
        B[ind2:ind2 + 4, ind] = (S2.T @ T1 @ S3).reshape(4, order='F')
        B[ind2:ind2 + 4, ind + 1] = (S2.T @ T2 @ S3).reshape(4, order='F')
        B[ind2:ind2 + 4, ind + 2:ind + 4] = np.kron((S3.T @ J3.T @ h1).reshape(2, 1), sw)
        B[ind2:ind2 + 4, ind + 4:ind + 6] = np.kron(sw, (S2.T @ K3 @ h1).reshape(2, 1))
    return f, Ap, B


def _gh_initial_obs(P1, P2, P3, x1, x2, x3):
    """ResslTFTPoseEstimation.m:72-81 (same in Nordberg :87-96, FaugPapa :58-67)."""
    N = x1.shape[1]
    points3D = triangulation3D([P1, P2, P3], np.vstack([x1, x2, x3]))
    p1 = P1 @ points3D
    p1 = p1[0:2, :] / p1[2:3, :]
    p2 = P2 @ points3D
    p2 = p2[0:2, :] / p2[2:3, :]
    p3 = P3 @ points3D
    p3 = p3[0:2, :] / p3[2:3, :]
    x = np.vstack([x1[0:2, :], x2[0:2, :], x3[0:2, :]]).reshape(6 * N, order='F')
    x_est = np.vstack([p1, p2, p3]).reshape(6 * N, order='F')
    return x, x_est


def _ressl_T(S, e21, e31, mn):
    T = np.zeros((3, 3, 3))
    for i in range(3):
        T[:, :, i] = (np.outer(S[:, i], e21) + np.outer(e31, mn[i, :])).T
    return T


def _ressl_constraintsGH(x, p, Ind):
    """ResslTFTPoseEstimation.m:110-177 (Ind is 0-based here)."""
    Ind2 = [k for k in range(3) if k != Ind]
    S = p[0:9].reshape(3, 3, order='F')
    e21 = np.ones(3)
    e21[Ind2] = p[9:11]
    e31 = p[17:20]
    mn = np.zeros((3, 3))
    mn[:, Ind2] = p[11:17].reshape(3, 2, order='F')
    T = _ressl_T(S, e21, e31, mn)
    g = np.array([np.sum(e31 ** 2) - 1, np.sum(S ** 2) - 1])
    C = np.zeros((2, 20))
    C[0, 17:20] = 2 * e31
    C[1, 0:9] = 2 * S.reshape(9, order='F')
    f, Ap, B = _trilinear_blocks(x, T)
    D = np.zeros((27, 20))
    D[:, 0:9] = np.kron(np.eye(3), np.kron(np.eye(3), e21.reshape(3, 1)))
    aux = np.zeros((3, 2))
    aux[Ind2, :] = np.eye(2)
    D[:, 9:11] = np.vstack([np.kron(S[:, i].reshape(3, 1), aux) for i in range(3)])
    D[:, 11:14] = np.kron(np.eye(3), np.kron(e31.reshape(3, 1), aux[:, 0:1]))
    D[:, 14:17] = np.kron(np.eye(3), np.kron(e31.reshape(3, 1), aux[:, 1:2]))
    D[:, 17:20] = np.vstack([np.kron(np.eye(3), mn[i, :].reshape(3, 1)) for i in range(3)])
    A = Ap @ D
    return f, g, A, B, C, np.zeros((2, 0))


def ResslTFTPoseEstimation(Corresp, CalM, return_debug=False):
    """TFT_methods/ResslTFTPoseEstimation.m:47-105"""
    x1, Normal1 = Normalize2Ddata(Corresp[0:2, :])
    x2, Normal2 = Normalize2Ddata(Corresp[2:4, :])
    x3, Normal3 = Normalize2Ddata(Corresp[4:6, :])
    T, P1, P2, P3 = linearTFT(x1, x2, x3)
    e21 = P2[:, 3].copy()
    Ind = int(np.argmax(np.abs(e21)))                   # first maximum, as MATLAB max
    e21 = e21 / e21[Ind]
    e31 = P3[:, 3].copy()
    e31 = e31 / np.linalg.norm(e31)
    S = np.stack([T[Ind, :, 0], T[Ind, :, 1], T[Ind, :, 2]], axis=1)
    aux = np.linalg.norm(S.reshape(9))
    S = S / aux
    T = T / aux
    Ind2 = [k for k in range(3) if k != Ind]
    mn = np.stack([e31 @ (T[:, :, i].T - np.outer(S[:, i], e21)) for i in range(3)], axis=0)
    mn = mn[:, Ind2]
    x, x_est = _gh_initial_obs(P1, P2, P3, x1, x2, x3)
    p = np.concatenate([S.reshape(9, order='F'), e21[Ind2], mn.reshape(6, order='F'), e31])
    func = lambda a, b, c: _ressl_constraintsGH(a, b, Ind)
    _, p_opt, _, it, reason = Gauss_Helmert(func, x_est, p, np.zeros(0), x, None, True)
    S = p_opt[0:9].reshape(3, 3, order='F')
    e21 = np.ones(3)
    e21[Ind2] = p_opt[9:11]
    mn = np.zeros((3, 3))
    mn[:, Ind2] = p_opt[11:17].reshape(3, 2, order='F')
    e31 = p_opt[17:20]
    T = _ressl_T(S, e21, e31, mn)
    T = transform_TFT(T, Normal1, Normal2, Normal3, 1)
    R_t_2, R_t_3 = R_t_from_TFT(T, CalM, Corresp)
    Reconst = _final_reconst(CalM, R_t_2, R_t_3, Corresp)
    if return_debug:
        return R_t_2, R_t_3, Reconst, T, it, dict(reason=reason, Ind=Ind, p0=p, p_opt=p_opt)
    return R_t_2, R_t_3, Reconst, T, it


# ---- Nordberg -------------------------------------------------------------
_NORD_IND = np.array([1, 7, 10, 12, 16, 19, 20, 21, 22, 25]) - 1   # NordbergTFT...m:82 (1-based linear, col-major)


def _transf_t(T0, U, V, W):
    """NordbergTFTPoseEstimation.m:217-222"""
    T = np.zeros((3, 3, 3))
    for i in range(3):
        T[:, :, i] = V.T @ (U[0, i] * T0[:, :, 0] + U[1, i] * T0[:, :, 1] + U[2, i] * T0[:, :, 2]) @ W
    return T


def _rodrigues(o, vec):
    cm = crossM(vec)
    return np.eye(3) + np.sin(o) * cm + (1 - np.cos(o)) * (cm @ cm)


def _nordberg_constrGH(obs, x):
    """NordbergTFTPoseEstimation.m:128-213"""
    o, vec, Rm = [], [], []
    for k in range(3):
        ok = np.linalg.norm(x[3 * k:3 * k + 3])
        vk = x[3 * k:3 * k + 3] / ok
        o.append(ok)
        vec.append(vk)
        Rm.append(_rodrigues(ok, vk))
    U, V, W = Rm
    paramT = x[9:19]
    tsv = np.zeros(27)
    tsv[_NORD_IND] = paramT
    Ts = _unvecT(tsv)
    T = _transf_t(Ts, U.T, V.T, W.T)
    f, Ap, B = _trilinear_blocks(obs, T)
    J = np.zeros((27, 19))
    for i in range(10):
        e = np.zeros(27)
        e[_NORD_IND[i]] = 1
        J[:, i + 9] = _vecT(_transf_t(_unvecT(e), U.T, V.T, W.T))
    e3 = np.eye(3)
    dR = [[None] * 3 for _ in range(3)]
    for k in range(3):
        ok, vk = o[k], vec[k]
        cm = crossM(vk)
        for i in range(3):
            dR[k][i] = (-vk[i] * np.sin(ok) * np.eye(3) + vk[i] * np.cos(ok) * cm
                        + np.sin(ok) * (1 / ok) * (crossM(e3[:, i]) - vk[i] * cm)
                        + vk[i] * np.sin(ok) * np.outer(vk, vk)
                        + (1 - np.cos(ok)) * (1 / ok) * (np.outer(vk, e3[i, :]) + np.outer(e3[:, i], vk)
                                                         - 2 * vk[i] * np.outer(vk, vk)))
    for i in range(3):
        J[:, i] = _vecT(_transf_t(Ts, dR[0][i].T, V.T, W.T))
        J[:, i + 3] = _vecT(_transf_t(Ts, U.T, dR[1][i].T, W.T))
        J[:, i + 6] = _vecT(_transf_t(Ts, U.T, V.T, dR[2][i].T))
    A = Ap @ J
    g = np.array([np.sum(paramT ** 2) - 1])
    C = np.zeros((1, 19))
    C[0, 9:19] = 2 * paramT
    return f, g, A, B, C, np.zeros((1, 0))


def _axis_angle(U):
    """NordbergTFTPoseEstimation.m:73-74"""
    v = _svdV(U - np.eye(3))
    vec = v[:, 2]
    o = np.arctan2(vec @ np.array([U[2, 1] - U[1, 2], U[0, 2] - U[2, 0], U[1, 0] - U[0, 1]]) / 2,
                   (np.trace(U) - 1) / 2)
    return vec, o


def nordberg_param0(T, P1, P2, P3, null_sign=1.0):
    """NordbergTFTPoseEstimation.m:55-94: projective fix-up for a rank-deficient P3(:,1:3) / P2(:,1:3), then the initial 19
    parameters (three axis-angle vectors and the 10 sparse tensor entries).  Returns the (possibly transformed) cameras and param0.
    null_sign: the sign of null(.) is whatever svd returns (MATLAB leaves it open); -1 replays the other choice, which gives
    projectively equivalent cameras and a different but equivalent parameter vector."""
    H = np.eye(4)
    if rank(P3[:, 0:3]) < 3:
        H[3, 0:3] = null_sign * null(P3[:, 0:3])[:, 0]
    elif rank(P2[:, 0:3]) < 3:
        H[3, 0:3] = null_sign * null(P2[:, 0:3])[:, 0]
    P1, P2, P3 = P1 @ H, P2 @ H, P3 @ H
    A = P2[:, 0:3]
    a = P2[:, 3]
    r = np.linalg.solve(A, a)
    Bm = P3[:, 0:3]
    b = P3[:, 3]
    s = np.linalg.solve(Bm, b)

    def orth(M):
        M = M @ _mpower_invsqrt(M.T @ M)
        return _sign(np.linalg.det(M)) * M
    cr, ca, cb = crossM(r), crossM(a), crossM(b)
    U = orth(np.stack([r, cr @ cr @ s, cr @ s], axis=1))
    V = orth(np.stack([a, ca @ A @ s, ca @ ca @ A @ s], axis=1))
    W = orth(np.stack([b, cb @ Bm @ r, cb @ cb @ Bm @ r], axis=1))
    vec_u, o_u = _axis_angle(U)
    vec_v, o_v = _axis_angle(V)
    vec_w, o_w = _axis_angle(W)
    Ts = _transf_t(T, U, V, W)
    paramT = _vecT(Ts)[_NORD_IND]
    paramT = paramT / np.linalg.norm(paramT)
    return P1, P2, P3, np.concatenate([vec_u * o_u, vec_v * o_v, vec_w * o_w, paramT])


def NordbergTFTPoseEstimation(Corresp, CalM, return_debug=False):
    """TFT_methods/NordbergTFTPoseEstimation.m:47-124"""
    x1, Normal1 = Normalize2Ddata(Corresp[0:2, :])
    x2, Normal2 = Normalize2Ddata(Corresp[2:4, :])
    x3, Normal3 = Normalize2Ddata(Corresp[4:6, :])
    T, P1, P2, P3 = linearTFT(x1, x2, x3)
    P1, P2, P3, param0 = nordberg_param0(T, P1, P2, P3)
    obs, obs_est = _gh_initial_obs(P1, P2, P3, x1, x2, x3)
    func = lambda a_, b_, c_: _nordberg_constrGH(a_, b_)
    _, param, _, it, reason = Gauss_Helmert(func, obs_est, param0, np.zeros(0), obs, None, True)
    Rm = []
    for k in range(3):
        ok = np.linalg.norm(param[3 * k:3 * k + 3])
        Rm.append(_rodrigues(ok, param[3 * k:3 * k + 3] / ok))
    U, V, W = Rm
    tsv = np.zeros(27)
    tsv[_NORD_IND] = param[9:19]
    T = _transf_t(_unvecT(tsv), U.T, V.T, W.T)
    T = transform_TFT(T, Normal1, Normal2, Normal3, 1)
    R_t_2, R_t_3 = R_t_from_TFT(T, CalM, Corresp)
    Reconst = _final_reconst(CalM, R_t_2, R_t_3, Corresp)
    if return_debug:
        return R_t_2, R_t_3, Reconst, T, it, dict(reason=reason, param0=param0, param=param)
    return R_t_2, R_t_3, Reconst, T, it


# ---- Faugeras-Papadopoulo -------------------------------------------------
def _minor(A, i, j):
    """FaugPapaTFTPoseEstimation.m:156-159 (signed cofactor; i, j 0-based here)."""
    h, w = A.shape
    rows = [r for r in range(h) if r != i]
    cols = [c for c in range(w) if c != j]
    return np.linalg.det(A[np.ix_(rows, cols)]) * (-1) ** (i + j)


def _fp_stack(T, idx3):
    """reshape([T(a1,b1,:),T(a2,b2,:),T(a3,b3,:)],3,3) (FaugPapa :132-135):
    horzcat of three 1x1x3 arrays is 1x3x3; column-major reshape to 3x3 gives
    element (c, i) = T(a_c, b_c, i)."""
    return np.stack([T[a, b, :] for (a, b) in idx3], axis=0)


def _faugpapa_constrGH(obs, x):
    """FaugPapaTFTPoseEstimation.m:87-153"""
    T = _unvecT(x)
    f, A, B = _trilinear_blocks(obs, T)
    g = np.zeros(12)
    C = np.zeros((12, 27))
    for i in range(3):
        g[i] = np.linalg.det(T[:, :, i])
        for j in range(3):
            for k in range(3):
                C[i, j + 3 * k + 9 * i] = _minor(T[:, :, i], j, k)
    i = -1
    for k2 in range(2):
        for k3 in range(2):
            for l2 in range(k2 + 1, 3):
                for l3 in range(k3 + 1, 3):
                    i += 1
                    A1 = _fp_stack(T, [(k2, k3), (k2, l3), (l2, l3)])
                    A2 = _fp_stack(T, [(k2, k3), (l2, k3), (l2, l3)])
                    A3 = _fp_stack(T, [(l2, k3), (k2, l3), (l2, l3)])
                    A4 = _fp_stack(T, [(k2, k3), (l2, k3), (k2, l3)])
                    d1, d2, d3, d4 = (np.linalg.det(A1), np.linalg.det(A2),
                                      np.linalg.det(A3), np.linalg.det(A4))
                    g[3 + i] = d1 * d2 - d3 * d4
                    for i1 in range(3):
                        C[3 + i, k2 + 3 * k3 + 9 * i1] = (_minor(A1, i1, 0) * d2 + d1 * _minor(A2, i1, 0)
                                                          - d3 * _minor(A4, i1, 0))
                        C[3 + i, k2 + 3 * l3 + 9 * i1] = (_minor(A1, i1, 1) * d2 - _minor(A3, i1, 1) * d4
                                                          - d3 * _minor(A4, i1, 2))
                        C[3 + i, l2 + 3 * l3 + 9 * i1] = (_minor(A1, i1, 2) * d2 + d1 * _minor(A2, i1, 2)
                                                          - _minor(A3, i1, 2) * d4)
                        C[3 + i, l2 + 3 * k3 + 9 * i1] = (d1 * _minor(A2, i1, 1) - _minor(A3, i1, 0) * d4
                                                          - d3 * _minor(A4, i1, 1))
    return f, g, A, B, C, np.zeros((12, 0))


def FaugPapaTFTPoseEstimation(Corresp, CalM, return_debug=False):
    """TFT_methods/FaugPapaTFTPoseEstimation.m:48-82"""
    x1, Normal1 = Normalize2Ddata(Corresp[0:2, :])
    x2, Normal2 = Normalize2Ddata(Corresp[2:4, :])
    x3, Normal3 = Normalize2Ddata(Corresp[4:6, :])
    T, P1, P2, P3 = linearTFT(x1, x2, x3)
    obs, obs_est = _gh_initial_obs(P1, P2, P3, x1, x2, x3)
    param0 = _vecT(T)
    func = lambda a_, b_, c_: _faugpapa_constrGH(a_, b_)
    _, param, _, it, reason = Gauss_Helmert(func, obs_est, param0, np.zeros(0), obs, None, True)
    T = _unvecT(param)
    T = transform_TFT(T, Normal1, Normal2, Normal3, 1)
    R_t_2, R_t_3 = R_t_from_TFT(T, CalM, Corresp)
    Reconst = _final_reconst(CalM, R_t_2, R_t_3, Corresp)
    if return_debug:
        return R_t_2, R_t_3, Reconst, T, it, dict(reason=reason, param0=param0, param=param)
    return R_t_2, R_t_3, Reconst, T, it


# --------------------------------------------------------------------------
# F_methods/optimF.m, F_methods/OptimFPoseEstimation.m   
# --------------------------------------------------------------------------
def _constraintsGH_F(x, p):
    """optimF.m:83-109"""
    N = x.shape[0] // 4
    xr = x.reshape(4, N, order='F')
    F = p.reshape(3, 3, order='F')
    Fl = p                                    # F(k), 1-based linear index k -> Fl[k-1]
    g = np.array([np.linalg.det(F), np.sum(Fl ** 2) - 1])
    C = np.array([[Fl[4] * Fl[8] - Fl[5] * Fl[7], Fl[5] * Fl[6] - Fl[3] * Fl[8], Fl[3] * Fl[7] - Fl[4] * Fl[6],
                   Fl[2] * Fl[7] - Fl[1] * Fl[8], Fl[0] * Fl[8] - Fl[2] * Fl[6], Fl[1] * Fl[6] - Fl[0] * Fl[7],
                   Fl[1] * Fl[5] - Fl[2] * Fl[4], Fl[2] * Fl[3] - Fl[0] * Fl[5], Fl[0] * Fl[4] - Fl[1] * Fl[3]],
                  2 * Fl])
    f = np.zeros(N); A = np.zeros((N, 9)); B = np.zeros((N, 4 * N))
    for i in range(N):
        x1 = np.array([xr[0, i], xr[1, i], 1.0]); x2 = np.array([xr[2, i], xr[3, i], 1.0])
        f[i] = x2 @ F @ x1
        A[i, :] = [x1[0] * x2[0], x1[0] * x2[1], x1[0], x1[1] * x2[0], x1[1] * x2[1], x1[1], x2[0], x2[1], 1]
        B[i, 4 * i:4 * i + 4] = [Fl[2] + Fl[0] * x2[0] + Fl[1] * x2[1], Fl[5] + Fl[3] * x2[0] + Fl[4] * x2[1],
                                 Fl[6] + Fl[0] * x1[0] + Fl[3] * x1[1], Fl[7] + Fl[1] * x1[0] + Fl[4] * x1[1]]
    return f, g, A, B, C, np.zeros((2, 0))


def optimF(p1, p2, return_debug=False):
    """F_methods/optimF.m:34-77"""
    N = p1.shape[1]
    if N != p2.shape[1] or N < 8:
        raise ValueError('At least 8 correspondences are necessary to compute the fundamental matrix linearly')
    if p1.shape[0] == 3:
        p1 = p1[0:2, :] / p1[2:3, :]
        p2 = p2[0:2, :] / p2[2:3, :]
    x1, Normal1 = Normalize2Ddata(p1)
    x2, Normal2 = Normalize2Ddata(p2)
    F = linearF(x1, x2)
    F = F / np.sqrt(np.sum(F.reshape(9, order='F') ** 2))                  # :50
    U, _, _ = _svd(F)
    epi21 = U[:, 2]                                                        # :53
    P1 = np.eye(3, 4)
    P2 = np.hstack([crossM(epi21) @ F, epi21.reshape(3, 1)])               # :55
    points3D = triangulation3D([P1, P2], np.vstack([x1, x2]))
    p1_est = P1 @ points3D; p1_est = p1_est[0:2, :] / p1_est[2:3, :]
    p2_est = P2 @ points3D; p2_est = p2_est[0:2, :] / p2_est[2:3, :]
    p = F.reshape(9, order='F')
    x = np.vstack([x1[0:2, :], x2[0:2, :]]).reshape(4 * N, order='F')
    x_est = np.vstack([p1_est, p2_est]).reshape(4 * N, order='F')
    _, p_opt, _, it, reason = Gauss_Helmert(lambda a, b, c: _constraintsGH_F(a, b), x_est, p, np.zeros(0), x, None, True)
    F = p_opt.reshape(3, 3, order='F')
    F = Normal2.T @ F @ Normal1                                            # :72
    U, D, V = _svd(F)
    D = D.copy(); D[2] = 0
    F = U @ np.diag(D) @ V.T                                               # :75-76
    if return_debug:
        return F, it, dict(reason=reason, p0=p, p_opt=p_opt)
    return F, it


def OptimFPoseEstimation(Corresp, CalM):
    """F_methods/OptimFPoseEstimation.m:44-73"""
    K1, K2, K3 = CalM[0:3, :], CalM[3:6, :], CalM[6:9, :]
    F21, it1 = optimF(Corresp[0:2, :], Corresp[2:4, :])
    F31, it2 = optimF(Corresp[0:2, :], Corresp[4:6, :])
    it = it1 + it2
    P1cam = np.hstack([K1, np.zeros((3, 1))])
    R2, t2 = _recover_R_t_core(K2.T @ F21 @ K1, P1cam, K2, Corresp[0:2, :], Corresp[2:4, :])
    R3, t3 = _recover_R_t_core(K3.T @ F31 @ K1, P1cam, K3, Corresp[0:2, :], Corresp[4:6, :])
    lam = _t3_scale(K1, K2, K3, R2, t2, R3, t3, Corresp)
    t3 = lam * t3
    R_t_2 = np.hstack([R2, t2.reshape(3, 1)])
    R_t_3 = np.hstack([R3, t3.reshape(3, 1)])
    Reconst = _final_reconst(CalM, R_t_2, R_t_3, Corresp)
    T = TFT_from_P(K1 @ np.eye(3, 4), K2 @ R_t_2, K3 @ R_t_3)
    return R_t_2, R_t_3, Reconst, T, it


# ---- Ponce-Hebert Pi-matrix methods -------------------
def _pi_constraintsGH(x, pi):
    """TFT_methods/PiPoseEstimation.m:109-182 (3 epipolar equations + 1 trilinearity per point, 9 constraints)."""
    N = x.shape[0] // 6
    pi21, pi31, pi41 = pi[0:3], pi[3:6], pi[6:9]
    pi12, pi32, pi42 = pi[9:12], pi[12:15], pi[15:18]
    pi13, pi23, pi43 = pi[18:21], pi[21:24], pi[24:27]
    F12 = np.outer(pi41, pi32) - np.outer(pi31, pi42)                       # :118-120
    F13 = np.outer(pi41, pi23) - np.outer(pi21, pi43)
    F23 = np.outer(pi42, pi13) - np.outer(pi12, pi43)
    g = np.array([pi41 @ pi41 - 1, pi42 @ pi42 - 1, pi43 @ pi43 - 1,
                  pi21 @ pi21 - 1, pi32 @ pi32 - 1, pi13 @ pi13 - 1,
                  pi21 @ pi41, pi32 @ pi42, pi13 @ pi43])                   # :123-125
    C = np.zeros((9, 27))                                                   # :128-137
    C[0, 6:9] = 2 * pi41
    C[1, 15:18] = 2 * pi42
    C[2, 24:27] = 2 * pi43
    C[3, 0:3] = 2 * pi21
    C[4, 12:15] = 2 * pi32
    C[5, 18:21] = 2 * pi13
    C[6, 0:3] = pi41; C[6, 6:9] = pi21
    C[7, 12:15] = pi42; C[7, 15:18] = pi32
    C[8, 18:21] = pi43; C[8, 24:27] = pi13
    f = np.zeros(4 * N)
    A = np.zeros((4 * N, 27))
    B = np.zeros((4 * N, 6 * N))
    for i in range(N):
        ind = 6 * i
        p1 = np.array([x[ind], x[ind + 1], 1.0])
        p2 = np.array([x[ind + 2], x[ind + 3], 1.0])
        p3 = np.array([x[ind + 4], x[ind + 5], 1.0])
        r = 4 * i
        a21, a31, a41 = pi21 @ p1, pi31 @ p1, pi41 @ p1
        a12, a32, a42 = pi12 @ p2, pi32 @ p2, pi42 @ p2
        a13, a23, a43 = pi13 @ p3, pi23 @ p3, pi43 @ p3
        f[r:r + 4] = [p1 @ F12 @ p2, p1 @ F13 @ p3, p2 @ F23 @ p3, a21 * a32 * a13 - a31 * a12 * a23]   # :152-153
        A[r, 3:9] = np.concatenate([-a42 * p1, a32 * p1])                   # :156-164
        A[r, 12:18] = np.concatenate([a41 * p2, -a31 * p2])
        A[r + 1, 0:3] = -a43 * p1; A[r + 1, 6:9] = a23 * p1
        A[r + 1, 21:27] = np.concatenate([a41 * p3, -a21 * p3])
        A[r + 2, 9:12] = -a43 * p2; A[r + 2, 15:18] = a13 * p2
        A[r + 2, 18:21] = a42 * p3; A[r + 2, 24:27] = -a12 * p3
        A[r + 3, 0:6] = np.concatenate([p1 * a32 * a13, -p1 * a12 * a23])
        A[r + 3, 9:15] = np.concatenate([-a31 * a23 * p2, a21 * a13 * p2])
        A[r + 3, 18:24] = np.concatenate([a21 * a32 * p3, -a31 * a12 * p3])
        B[r, ind:ind + 2] = (F12 @ p2)[0:2]; B[r, ind + 2:ind + 4] = (p1 @ F12)[0:2]            # :166-171
        B[r + 1, ind:ind + 2] = (F13 @ p3)[0:2]; B[r + 1, ind + 4:ind + 6] = (p1 @ F13)[0:2]
        B[r + 2, ind + 2:ind + 4] = (F23 @ p3)[0:2]; B[r + 2, ind + 4:ind + 6] = (p2 @ F23)[0:2]
        B[r + 3, ind:ind + 2] = (pi21 * a32 * a13 - pi31 * a12 * a23)[0:2]
        B[r + 3, ind + 2:ind + 4] = (pi32 * a21 * a13 - pi12 * a31 * a23)[0:2]
        B[r + 3, ind + 4:ind + 6] = (pi13 * a21 * a32 - pi23 * a31 * a12)[0:2]
    return f, g, A, B, C, np.zeros((9, 0))


def _pi_finish(P1, P2, P3, Normal1, Normal2, Normal3, CalM, Corresp):
    T = TFT_from_P(P1, P2, P3)
    T = transform_TFT(T, Normal1, Normal2, Normal3, 1)
    R_t_2, R_t_3 = R_t_from_TFT(T, CalM, Corresp)
    return R_t_2, R_t_3, _final_reconst(CalM, R_t_2, R_t_3, Corresp), T


def PiPoseEstimation(Corresp, CalM, return_debug=False, null=null, cam_signs=(1.0, 1.0), init_only=False):
    """TFT_methods/PiPoseEstimation.m:50-105.  `null` / `cam_signs` / `init_only`: see PiColPoseEstimation
    (here the sign choices only relabel an equivalent problem)."""
    inv = np.linalg.inv
    x1, Normal1 = Normalize2Ddata(Corresp[0:2, :])
    x2, Normal2 = Normalize2Ddata(Corresp[2:4, :])
    x3, Normal3 = Normalize2Ddata(Corresp[4:6, :])
    _, P1, P2, P3 = linearTFT(x1, x2, x3)
    P2, P3 = cam_signs[0] * P2, cam_signs[1] * P3
    M = np.hstack([null(P1), null(P2), null(P3)])                           # :61-63
    M = np.hstack([M, null(M.T)])
    P1, P2, P3 = P1 @ M, P2 @ M, P3 @ M
    Pi1 = inv(P1[:, 1:4]); Pi2 = inv(P2[:, [0, 2, 3]]); Pi3 = inv(P3[:, [0, 1, 3]])             # :66-69
    Pi1 = np.vstack([np.zeros(3), Pi1])
    Pi2 = np.vstack([Pi2[0], np.zeros(3), Pi2[1:3]])
    Pi3 = np.vstack([Pi3[0:2], np.zeros(3), Pi3[2]])
    Pi1 = Pi1 / np.linalg.norm(Pi1[3]); Pi2 = Pi2 / np.linalg.norm(Pi2[3]); Pi3 = Pi3 / np.linalg.norm(Pi3[3])   # :72
    Q = np.eye(4)                                                           # :73-76
    Q[0, 0] = 1.0 / np.linalg.norm(Pi3[0] - (Pi3[0] @ Pi3[3]) * Pi3[3]); Q[0, 3] = -Q[0, 0] * (Pi3[0] @ Pi3[3])
    Q[1, 1] = 1.0 / np.linalg.norm(Pi1[1] - (Pi1[1] @ Pi1[3]) * Pi1[3]); Q[1, 3] = -Q[1, 1] * (Pi1[1] @ Pi1[3])
    Q[2, 2] = 1.0 / np.linalg.norm(Pi2[2] - (Pi2[2] @ Pi2[3]) * Pi2[3]); Q[2, 3] = -Q[2, 2] * (Pi2[2] @ Pi2[3])
    Pi1, Pi2, Pi3 = Q @ Pi1, Q @ Pi2, Q @ Pi3
    Qi = inv(Q)
    P1, P2, P3 = P1 @ Qi, P2 @ Qi, P3 @ Qi                                  # :79
    x, x_est = _gh_initial_obs(P1, P2, P3, x1, x2, x3)                      # :80-83, :87-88
    pi = np.concatenate([Pi1[1:4].reshape(9), Pi2[[0, 2, 3]].reshape(9), Pi3[[0, 1, 3]].reshape(9)])   # :86 (rows, via .')
    if init_only:
        return pi, x_est
    func = lambda a, b, c: _pi_constraintsGH(a, b)
    _, pi_opt, _, it, reason = Gauss_Helmert(func, x_est, pi, np.zeros(0), x, None, True)
    Pi1 = pi_opt[0:9].reshape(3, 3); Pi2 = pi_opt[9:18].reshape(3, 3); Pi3 = pi_opt[18:27].reshape(3, 3)   # :94-96
    P1 = np.zeros((3, 4)); P2 = np.zeros((3, 4)); P3 = np.zeros((3, 4))
    P1[:, 1:4] = inv(Pi1)
    P2[:, [0, 2, 3]] = inv(Pi2)
    P3[:, [0, 1, 3]] = inv(Pi3)
    R_t_2, R_t_3, Reconst, T = _pi_finish(P1, P2, P3, Normal1, Normal2, Normal3, CalM, Corresp)
    if return_debug:
        return R_t_2, R_t_3, Reconst, T, it, dict(reason=reason, p0=pi, p_opt=pi_opt, x_est=x_est)
    return R_t_2, R_t_3, Reconst, T, it


def _picol_constraintsGH(x, pi):
    """TFT_methods/PiColPoseEstimation.m:134-218 (3 epipolar equations + 2 trilinearities per point, 11 constraints).
    Restated literally, including the sign of A(ind2+4,1:3) (:176), which is not the derivative of f(ind2+4)."""
    N = x.shape[0] // 6
    pi21, pi31, pi41 = pi[0:3], pi[3:6], pi[6:9]
    pi12, pi32, pi42 = pi[9:12], pi[12:15], pi[15:18]
    w3, pi33, pi43 = pi[18:21], pi[21:24], pi[24:27]
    F12 = np.outer(pi41, pi32) - np.outer(pi31, pi42)                       # :144-146
    F13 = np.outer(pi41, pi33) - np.outer(pi31, pi43)
    F23 = np.outer(pi42, pi33) - np.outer(pi32, pi43)
    g = np.array([pi21 @ pi21 - 1, pi12 @ pi12 - 1, w3 @ w3 - 1, pi33 @ pi33 - 1, pi43 @ pi43 - 1,
                  pi21 @ pi31, pi21 @ pi41, pi31 @ pi41, pi12 @ pi32, pi12 @ pi42, pi32 @ pi42])   # :149-152
    C = np.zeros((11, 27))                                                  # :155-160
    C[0, 0:3] = 2 * pi21; C[1, 9:12] = 2 * pi12
    C[2, 18:21] = 2 * w3; C[3, 21:24] = 2 * pi33; C[4, 24:27] = 2 * pi43
    C[5, 0:6] = np.concatenate([pi31, pi21]); C[8, 9:15] = np.concatenate([pi32, pi12])
    C[6, 0:3] = pi41; C[6, 6:9] = pi21; C[9, 9:12] = pi42; C[9, 15:18] = pi12
    C[7, 3:9] = np.concatenate([pi41, pi31]); C[10, 12:18] = np.concatenate([pi42, pi32])
    f = np.zeros(5 * N)
    A = np.zeros((5 * N, 27))
    B = np.zeros((5 * N, 6 * N))
    for i in range(N):
        ind = 6 * i
        p1 = np.array([x[ind], x[ind + 1], 1.0])
        p2 = np.array([x[ind + 2], x[ind + 3], 1.0])
        p3 = np.array([x[ind + 4], x[ind + 5], 1.0])
        r = 5 * i
        a21, a31, a41 = pi21 @ p1, pi31 @ p1, pi41 @ p1
        a12, a32, a42 = pi12 @ p2, pi32 @ p2, pi42 @ p2
        aw3, a33, a43 = w3 @ p3, pi33 @ p3, pi43 @ p3
        f[r:r + 5] = [p1 @ F12 @ p2, p1 @ F13 @ p3, p2 @ F23 @ p3,
                      a31 * a32 * aw3 + (a31 * a12 - a21 * a32) * a33,
                      a41 * a42 * aw3 + (a41 * a12 - a21 * a42) * a43]      # :175-177
        A[r, 3:9] = np.concatenate([-a42 * p1, a32 * p1]); A[r, 12:18] = np.concatenate([a41 * p2, -a31 * p2])       # :180-181
        A[r + 1, 3:9] = np.concatenate([-a43 * p1, a33 * p1]); A[r + 1, 21:27] = np.concatenate([a41 * p3, -a31 * p3])   # :182-183
        A[r + 2, 12:18] = np.concatenate([-a43 * p2, a33 * p2]); A[r + 2, 21:27] = np.concatenate([a42 * p3, -a32 * p3])  # :184-185
        A[r + 3, 0:6] = np.concatenate([p1 * a32 * a33, p1 * (a32 * aw3 + a12 * a33)])                                 # :186-189
        A[r + 3, 9:15] = np.concatenate([p2 * a31 * a33, p2 * (a31 * aw3 - a21 * a33)])
        A[r + 3, 18:24] = np.concatenate([p3 * a31 * a32, p3 * (a31 * a12 - a21 * a32)])
        A[r + 4, 0:3] = -p1 * a42 * a43; A[r + 4, 6:9] = p1 * (a42 * aw3 + a12 * a43)                                   # :190-193
        A[r + 4, 9:12] = p2 * a41 * a43; A[r + 4, 15:18] = p2 * (a41 * aw3 - a21 * a43)
        A[r + 4, 18:21] = p3 * a41 * a42; A[r + 4, 24:27] = p3 * (a41 * a12 - a21 * a42)
        B[r, ind:ind + 2] = (F12 @ p2)[0:2]; B[r, ind + 2:ind + 4] = (p1 @ F12)[0:2]                                    # :195-197
        B[r + 1, ind:ind + 2] = (F13 @ p3)[0:2]; B[r + 1, ind + 4:ind + 6] = (p1 @ F13)[0:2]
        B[r + 2, ind + 2:ind + 4] = (F23 @ p3)[0:2]; B[r + 2, ind + 4:ind + 6] = (p2 @ F23)[0:2]
        B[r + 3, ind:ind + 2] = (pi31 * (a32 * aw3 + a12 * a33) - pi21 * a32 * a33)[0:2]                              # :198-201
        B[r + 3, ind + 2:ind + 4] = (a31 * pi32 * aw3 + (a31 * pi12 - a21 * pi32) * a33)[0:2]
        B[r + 3, ind + 4:ind + 6] = (a31 * a32 * w3 + (a31 * a12 - a21 * a32) * pi33)[0:2]
        B[r + 4, ind:ind + 2] = (pi41 * a42 * aw3 + (pi41 * a12 - pi21 * a42) * a43)[0:2]                             # :202-205
        B[r + 4, ind + 2:ind + 4] = (a41 * pi42 * aw3 + (a41 * pi12 - a21 * pi42) * a43)[0:2]
        B[r + 4, ind + 4:ind + 6] = (a41 * a42 * w3 + (a41 * a12 - a21 * a42) * pi43)[0:2]
    return f, g, A, B, C, np.zeros((11, 0))


def PiColPoseEstimation(Corresp, CalM, return_debug=False, null=null, cam_signs=(1.0, 1.0), init_only=False):
    """TFT_methods/PiColPoseEstimation.m:50-129 (Ponce-Hebert parameterisation for collinear camera centres).

    The result depends on sign/basis choices the reference leaves to MATLAB's svd: the signs of the
    projective cameras P2, P3 of linearTFT (epipoles are sign-free), the sign of every null(P) and
    the basis of the two-dimensional null(M.') -- :93-94 divide by u1.'*B*v2 where covariance needs
    u2.'*B*v2, so these gauges do not cancel.  `null` and `cam_signs` expose them to the tests (the
    defaults are LAPACK's, as literal as a restatement can be); `init_only` returns (pi, x_est)."""
    inv = np.linalg.inv
    nrm = np.linalg.norm
    x1, Normal1 = Normalize2Ddata(Corresp[0:2, :])
    x2, Normal2 = Normalize2Ddata(Corresp[2:4, :])
    x3, Normal3 = Normalize2Ddata(Corresp[4:6, :])
    _, P1, P2, P3 = linearTFT(x1, x2, x3)
    P2, P3 = cam_signs[0] * P2, cam_signs[1] * P3
    M = np.hstack([null(P1), null(P2)])                                     # :61-63
    coeff = np.linalg.lstsq(M, null(P3), rcond=None)[0].ravel()             # M\null(P3): 4x2 least squares
    M = np.hstack([coeff[0] * M[:, 0:1], coeff[1] * M[:, 1:2], null(M.T)])
    P1, P2, P3 = P1 @ M, P2 @ M, P3 @ M
    Pi1 = inv(P1[:, 1:4]); Pi2 = inv(P2[:, [0, 2, 3]]); Pi3 = inv(P3[:, 1:4])                    # :66-69
    Pi1 = np.vstack([np.zeros(3), Pi1])
    Pi2 = np.vstack([Pi2[0], np.zeros(3), Pi2[1:3]])
    Pi3 = np.vstack([np.zeros(3), Pi3])
    Pi1 = Pi1 / nrm(Pi1[3]); Pi2 = Pi2 / nrm(Pi2[3]); Pi3 = Pi3 / nrm(Pi3[3])                    # :72
    Q1 = np.eye(4)
    u1, v1 = Pi1[2].copy(), Pi1[3].copy()
    u2, v2 = Pi2[2].copy(), Pi2[3].copy()
    tol = 1e-10                                                             # :79-89
    A = (v1 @ v1) * (u2 @ v2) - (u1 @ v1) * (v2 @ v2)
    B = (v1 @ v1) * (u2 @ u2) - (u1 @ u1) * (v2 @ v2)
    C = (u1 @ v1) * (u2 @ u2) - (u1 @ u1) * (u2 @ v2)
    if abs(A) > tol and (B * B - 4 * A * C) >= 0 and abs(C) > tol:
        Q1[2, 3] = (-B + np.sqrt(B * B - 4 * A * C)) / (2 * A)
        Q1[3, 2] = (-B + np.sqrt(B * B - 4 * A * C)) / (2 * C)
    else:
        raise ValueError('The minimal param could not be found')
    A = np.outer(u1, v1) - np.outer(v1, u1); B = np.outer(u2, v2) - np.outer(v2, u2)             # :90-94
    Q1[1, 3] = (Pi1[1] @ A @ u1) / (u1 @ A @ v1)
    Q1[1, 2] = (Pi1[1] @ A.T @ v1) / (u1 @ A @ v1)
    Q1[0, 3] = (Pi2[0] @ B @ u2) / (u1 @ B @ v2)
    Q1[0, 2] = (Pi2[0] @ B.T @ v2) / (u1 @ B @ v2)
    Pi1, Pi2, Pi3 = Q1 @ Pi1, Q1 @ Pi2, Q1 @ Pi3                           # :96-104
    Pi1 = Pi1 / nrm(Pi1[1])
    Pi2 = Pi2 / nrm(Pi2[0])
    Pi3 = Pi3 / nrm(Pi3[1] - Pi3[0])
    Q2 = np.eye(4)
    Q2[2, 2] = 1.0 / nrm(Pi3[2])
    Q2[3, 3] = 1.0 / nrm(Pi3[3])
    Pi1, Pi2, Pi3 = Q2 @ Pi1, Q2 @ Pi2, Q2 @ Pi3
    Pi3[0:2] = Pi3[0:2] - Pi3[[0, 0]]
    Qi = inv(Q2 @ Q1)                                                       # :106
    P1, P2, P3 = P1 @ Qi, P2 @ Qi, P3 @ Qi
    x, x_est = _gh_initial_obs(P1, P2, P3, x1, x2, x3)                      # :107-110, :114-115
    pi = np.concatenate([Pi1[1:4].reshape(9), Pi2[[0, 2, 3]].reshape(9), Pi3[1:4].reshape(9)])   # :113
    if init_only:
        return pi, x_est
    func = lambda a, b, c: _picol_constraintsGH(a, b)
    _, pi_opt, _, it, reason = Gauss_Helmert(func, x_est, pi, np.zeros(0), x, None, True)
    Pi1 = pi_opt[0:9].reshape(3, 3); Pi2 = pi_opt[9:18].reshape(3, 3); Pi3 = pi_opt[18:27].reshape(3, 3)   # :121-123
    P1 = np.zeros((3, 4)); P2 = np.zeros((3, 4)); P3 = np.zeros((3, 4))
    P1[:, 1:4] = inv(Pi1)
    P2[:, [0, 2, 3]] = inv(Pi2)
    P3[:, 1:4] = inv(Pi3); P3[:, 0] = -P3[:, 1]                             # :127
    R_t_2, R_t_3, Reconst, T = _pi_finish(P1, P2, P3, Normal1, Normal2, Normal3, CalM, Corresp)
    if return_debug:
        return R_t_2, R_t_3, Reconst, T, it, dict(reason=reason, p0=pi, p_opt=pi_opt, x_est=x_est)
    return R_t_2, R_t_3, Reconst, T, it

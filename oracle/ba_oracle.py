"""
TEST INFRASTRUCTURE ONLY -- CPU restatement of Optimization/BundleAdjustment.m.

PARITY UNPINNED, twice over: the reference cannot run here (MATLAB), and its optimiser is MATLAB's closed-source
`lsqnonlin(..., 'Algorithm','levenberg-marquardt')` (BundleAdjustment.m:101-103).  What IS restated literally:

  * the pre-processing: per-view Normalize2Ddata folded into the calibration (:52-56) -- NOTE what it does to a view with a
    missing observation: `mean` over the view's points (Normalize2Ddata.m:34-35) is NaN, so EVERY point of that view and its
    calibration become NaN, and the callback's `isnan` test (:165) then skips the whole view --, optional initial
    triangulation over the views that see the point (:59-77), change of coordinates to camera 1 (:80-86), the three Euler angles of each rotation
    (:89-96), the variable vector [angles(:,2:M), translations(:,2:M), Reconst] (:100);
  * the residual / Jacobian callback `bundleadjustment_LM` (:128-204): observed minus projected point, analytic
    derivatives through Gamma and the Rx*Ry*Rz parameterisation;
  * the post-processing: R = Rx*Ry*Rz, scale 1/|t2|, `repr_err = norm(func(variables))` in NORMALISED coordinates (:105-123).

The optimiser itself is a plain Levenberg-Marquardt with the documented defaults of lsqnonlin's LM (InitDamping 0.01,
damping x10 / /10, FunctionTolerance = StepTolerance = 1e-6, 400 iterations, no scaling); the HIP kernel follows THIS
loop statement by statement, and both are checked against an independent optimiser (scipy's MINPACK lmder) at the
converged optimum -- the only level at which the reference itself could be matched.
"""
import numpy as np

from . import tft_oracle as O

LM_INIT_DAMPING = 0.01
LM_TOL_FUN = 1e-6
LM_TOL_X = 1e-6
LM_MAX_ITER = 400


def _rot_parts(a):
    cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Dx = np.array([[0, 0, 0], [0, -sx, -cx], [0, cx, -sx]])
    Dy = np.array([[-sy, 0, cy], [0, 0, 0], [-cy, 0, -sy]])
    Dz = np.array([[-sz, -cz, 0], [cz, -sz, 0], [0, 0, 0]])
    return Rx, Ry, Rz, Dx, Dy, Dz


def bundleadjustment_LM(variables, Corresp, CalM, want_jacobian=True):
    """BundleAdjustment.m:128-204, with the `isnan` branch of :165-167 (the rows of a skipped observation stay zero)."""
    N = Corresp.shape[1]
    M = Corresp.shape[0] // 2
    v = variables.reshape(3, N + 2 * (M - 1), order='F')
    angles, translations, pts = v[:, 0:M - 1], v[:, M - 1:2 * (M - 1)], v[:, 2 * (M - 1):]
    cams = []
    for j in range(M):
        K = CalM[3 * j:3 * j + 3]
        if j == 0:
            cams.append((K, K @ np.eye(3, 4), None))
        else:
            parts = _rot_parts(angles[:, j - 1])
            Rx, Ry, Rz = parts[0:3]
            cams.append((K, K @ np.hstack([Rx @ Ry @ Rz, translations[:, j - 1:j]]), parts))
    f = np.zeros(2 * M * N)
    J = np.zeros((2 * M * N, 3 * (2 * (M - 1) + N))) if want_jacobian else None
    for i in range(N):
        X = pts[:, i]
        for j in range(M):
            if np.isnan(Corresp[2 * j, i]):                                              # :165-167
                continue
            K, P, parts = cams[j]
            ind = 2 * M * i + 2 * j
            vv = P @ np.append(X, 1.0)
            f[ind:ind + 2] = Corresp[2 * j:2 * j + 2, i] - vv[0:2] / vv[2]          # Dist(point, Gamma(P*[Point;1]))
            if not want_jacobian:
                continue
            dgamma = np.array([[1 / vv[2], 0, -vv[0] / vv[2] ** 2], [0, 1 / vv[2], -vv[1] / vv[2] ** 2]])
            Jac = np.zeros((3, J.shape[1]))
            Jac[:, 6 * (M - 1) + 3 * i:6 * (M - 1) + 3 * i + 3] = P[:, 0:3]
            if j > 0:
                Rx, Ry, Rz, Dx, Dy, Dz = parts
                Jac[:, 3 * (M - 1) + 3 * (j - 1):3 * (M - 1) + 3 * j] = K
                Jac[:, 3 * (j - 1):3 * j] = np.stack([K @ (Dx @ Ry @ Rz) @ X, K @ (Rx @ Dy @ Rz) @ X, K @ (Rx @ Ry @ Dz) @ X], axis=1)
            J[ind:ind + 2, :] = -dgamma @ Jac                                          # dydist = -eye(2)
    return (f, J) if want_jacobian else f


def levenberg_marquardt(func, x0):
    """The LM loop shared with the HIP kernel (see the module header).  Returns x, successful iterations, evaluations."""
    x = x0.copy()
    lam = LM_INIT_DAMPING
    F, J = func(x, True)
    S = float(F @ F)
    g = J.T @ F
    H = J.T @ J
    it, evals = 0, 1
    while it < LM_MAX_ITER:
        step = -np.linalg.solve(H + lam * np.eye(H.shape[0]), g)
        Ft = func(x + step, False)
        evals += 1
        St = float(Ft @ Ft)
        small_step = np.linalg.norm(step) < LM_TOL_X * (np.sqrt(np.finfo(float).eps) + np.linalg.norm(x))
        if St < S:
            x = x + step
            it += 1
            done = abs(St - S) <= LM_TOL_FUN * S or small_step
            F, J = func(x, True)
            S = float(F @ F)
            g = J.T @ F
            H = J.T @ J
            lam = lam / 10.0
            if done:
                break
        else:
            lam = lam * 10.0
            if small_step or lam > 1e16:
                break
    return x, it, evals


def _euler_angles(R):
    """BundleAdjustment.m:92-94"""
    return np.array([-np.arctan2(R[1, 2], R[2, 2]), -np.arctan2(-R[0, 2], np.linalg.norm(R[1:3, 2])), -np.arctan2(R[0, 1], R[0, 0])])


def BundleAdjustment(CalM, R_t_0, Corresp, Reconst0=None, return_debug=False):
    """Optimization/BundleAdjustment.m:49-124.  CalM 3Mx3, R_t_0 3Mx4, Corresp 2MxN, Reconst0 3xN or None."""
    CalM = CalM.copy(); Corresp = Corresp.copy(); R_t_0 = R_t_0.copy()
    M = Corresp.shape[0] // 2
    N = Corresp.shape[1]
    for j in range(M):                                                                   # :52-56
        new_Corr, Normal = O.Normalize2Ddata(Corresp[2 * j:2 * j + 2])
        Corresp[2 * j:2 * j + 2] = new_Corr
        CalM[3 * j:3 * j + 3] = Normal @ CalM[3 * j:3 * j + 3]
    if Reconst0 is None:                                                                 # :59-77
        if not np.isnan(Corresp).any():
            X = O.triangulation3D([CalM[3 * j:3 * j + 3] @ R_t_0[3 * j:3 * j + 3] for j in range(M)], Corresp)
            Reconst0 = X[0:3] / X[3:4]
        else:                                                                            # per point, the views that see it (:63-72)
            Reconst0 = np.zeros((3, N))
            for i in range(N):
                js = [j for j in range(M) if not np.isnan(Corresp[2 * j, i])]
                if len(js) < 2:
                    raise ValueError("triangulation3D.m:36-38 returns nothing for fewer than two cameras: BundleAdjustment.m:73-74 stops with an error")
                X = O.triangulation3D([CalM[3 * j:3 * j + 3] @ R_t_0[3 * j:3 * j + 3] for j in js], np.concatenate([Corresp[2 * j:2 * j + 2, i:i + 1] for j in js]))
                Reconst0[:, i] = X[0:3, 0] / X[3, 0]
    cc = R_t_0[0:3].copy()                                                               # :80-86
    R_t_0[0:3] = np.eye(3, 4)
    for j in range(1, M):
        R_t_0[3 * j:3 * j + 3, 3] = R_t_0[3 * j:3 * j + 3, 3] - R_t_0[3 * j:3 * j + 3, 0:3] @ cc[:, 0:3].T @ cc[:, 3]
        R_t_0[3 * j:3 * j + 3, 0:3] = R_t_0[3 * j:3 * j + 3, 0:3] @ cc[:, 0:3].T
    Reconst0 = cc[:, 0:3] @ Reconst0 + cc[:, 3:4]
    angles0 = np.stack([_euler_angles(R_t_0[3 * j:3 * j + 3, 0:3]) for j in range(M)], axis=1)    # :89-96
    trans0 = np.stack([R_t_0[3 * j:3 * j + 3, 3] for j in range(M)], axis=1)
    x0 = np.hstack([angles0[:, 1:], trans0[:, 1:], Reconst0]).reshape(-1, order='F')            # :100
    func = lambda x, jac: bundleadjustment_LM(x, Corresp, CalM, jac)
    x, it, evals = levenberg_marquardt(func, x0)
    repr_err = float(np.linalg.norm(func(x, False)))                                             # :105
    v = x.reshape(3, N + 2 * (M - 1), order='F')
    angles, trans = v[:, 0:M - 1], v[:, M - 1:2 * (M - 1)]
    scale = 1.0 / np.linalg.norm(trans[:, 0])                                                    # :112
    R_t = np.zeros((3 * M, 4))
    R_t[0:3] = np.eye(3, 4)
    for j in range(M - 1):
        Rx, Ry, Rz = _rot_parts(angles[:, j])[0:3]
        R_t[3 * j + 3:3 * j + 6] = np.hstack([Rx @ Ry @ Rz, scale * trans[:, j:j + 1]])
    Reconst = scale * v[:, 2 * (M - 1):]
    if return_debug:
        return R_t, Reconst, it, repr_err, dict(x0=x0, x=x, evals=evals, Corresp_n=Corresp, CalM_n=CalM, cost0=float(np.linalg.norm(func(x0, False))))
    return R_t, Reconst, it, repr_err

"""
Synthetic three-view scenes with the geometry of the reference's generator
(auxiliar_functions/generateSyntheticScene.m:45-115), batched.

Host-side input generation for tests and bench.py.  MATLAB's rng/rand/randn
stream is not reproducible outside MATLAB, so scenes are drawn from numpy's
counter-based Philox generator keyed by `seed`; the *geometry* (calibration,
camera centres, look-at rotations, point cube, in-image rejection, ground
truth R_t) follows the reference line by line.
"""
import numpy as np


def _crossM(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=float)


def _rotation(u, v):
    # generateSyntheticScene.m:119-135
    u = u / np.linalg.norm(u)
    v = v / np.linalg.norm(v)
    w = np.cross(u, v)
    s = np.linalg.norm(w)
    c = float(np.dot(u, v))
    w = w / s
    return c * np.eye(3) + s * _crossM(w) + (1 - c) * np.outer(w, w)


def scene_cameras(focalL=50.0, angle=None):
    """K, (P1,P2,P3), ground-truth R_t = [R_t_2, R_t_3], CalM (9x3).
    generateSyntheticScene.m:45-72,113-115."""
    if angle is None or angle < 70 or angle > 180:
        p_coll = 0.0
    else:
        a = angle * np.pi / 180.0
        p_coll = 1 - np.sin(a) / (np.sqrt(2) * (np.cos(a) - 1))
    k = focalL / 50.0
    pix = 50.0
    K = np.array([[50 * k * pix, 0, 18 * pix], [0, 50 * k * pix, 12 * pix], [0, 0, 1]])
    C1 = k * np.array([0., -1400, 400]) + k * p_coll * np.array([0., 300, -300])
    C2 = k * np.array([-400., -1000, 0]) + k * p_coll * np.array([0., -100, 100])
    C3 = k * np.array([600., -800, -200]) + k * p_coll * np.array([0., -300, 300])
    down = np.array([0., 0, -1])
    R1, R2, R3 = _rotation(C1, down), _rotation(C2, down), _rotation(C3, down)
    Ps = []
    for R, C in ((R1, C1), (R2, C2), (R3, C3)):
        P = K @ R @ np.hstack([np.eye(3), -C.reshape(3, 1)])
        Ps.append(P * np.sqrt(24) / np.linalg.norm(P, 2))
    R_t_2 = R2 @ np.hstack([R1.T, (C1 - C2).reshape(3, 1)])
    R_t_3 = R3 @ np.hstack([R1.T, (C1 - C3).reshape(3, 1)])
    CalM = np.vstack([K, K, K])
    return K, Ps, [R_t_2, R_t_3], CalM


def generate_scene_batch(B, N, noise=1.0, seed=0, focalL=50.0, angle=None):
    """B independent scenes of N correspondences each.

    Returns
      Corresp  (B, N, 6) float64, C-contiguous: element [b, n, :] is
               [x1 y1 x2 y2 x3 y3] -- byte-identical to a MATLAB 6 x N x B
               column-major array (the layout of the C ABI);
      CalM     (9, 3) float64 (row-major numpy; use `calm_colmajor` for the ABI);
      R_t0     list of the two 3x4 ground-truth poses;
      points3D (B, N, 3).
    """
    K, Ps, R_t0, CalM = scene_cameras(focalL, angle)
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    pix = 50.0
    Corresp = np.empty((B, N, 6))
    points3D = np.empty((B, N, 3))
    filled = np.zeros(B, dtype=np.int64)
    todo = np.arange(B)
    Pst = np.stack(Ps)                                   # 3 x 3 x 4
    while todo.size:
        M = N + max(8, N // 8)                           # oversample; rejection is rare
        X = 400 * rng.random((todo.size, M, 3)) - 200    # :82
        Xh = np.concatenate([X, np.ones((todo.size, M, 1))], axis=2)
        x = np.einsum('vij,bmj->bmvi', Pst, Xh)          # b, m, view, 3
        x = x[..., 0:2] / x[..., 2:3]
        x = x + rng.standard_normal(x.shape) * noise     # :90-92
        x = x.reshape(todo.size, M, 6)
        xs, ys = x[..., 0::2], x[..., 1::2]
        inside = np.all((xs <= 36 * pix) & (ys <= 24 * pix) & (xs >= 0) & (ys >= 0), axis=2)  # :95-100
        still = []
        for r, b in enumerate(todo):
            idx = np.nonzero(inside[r])[0][: N - filled[b]]
            Corresp[b, filled[b]:filled[b] + idx.size] = x[r, idx]
            points3D[b, filled[b]:filled[b] + idx.size] = X[r, idx]
            filled[b] += idx.size
            if filled[b] < N:
                still.append(b)
        todo = np.array(still, dtype=np.int64)
    return Corresp, CalM, R_t0, points3D


def calm_colmajor(CalM):
    """9x3 numpy CalM -> the 27 doubles of a MATLAB column-major 9x3 array."""
    return np.ascontiguousarray(np.asarray(CalM, dtype=np.float64).T).reshape(27)


def generate_multiview_scene(M, N, noise=1.0, seed=0, focalL=50.0):
    """One scene of N points seen by M cameras (BundleAdjustment.m takes any number of views): the three cameras of
    generateSyntheticScene.m:52-60 first, further centres on the same side of the point cube, every camera looking at the origin.
    Returns Corresp (2M x N), CalM (3M x 3), R_t (3M x 4 ground truth, first pose [I|0]), points3D (3 x N)."""
    k = focalL / 50.0
    pix = 50.0
    K = np.array([[50 * k * pix, 0, 18 * pix], [0, 50 * k * pix, 12 * pix], [0, 0, 1]])
    centres = [np.array([0., -1400, 400]), np.array([-400., -1000, 0]), np.array([600., -800, -200]), np.array([300., -1200, 500]),
               np.array([-700., -900, -300]), np.array([100., -1500, -100])]
    if not 2 <= M <= len(centres):
        raise ValueError("2 .. %d views" % len(centres))
    down = np.array([0., 0, -1])
    Cs = [k * c for c in centres[:M]]
    Rs = [_rotation(c, down) for c in Cs]
    Ps = [K @ R @ np.hstack([np.eye(3), -c.reshape(3, 1)]) for R, c in zip(Rs, Cs)]
    rng = np.random.Generator(np.random.Philox(key=int(seed)))
    X = np.empty((3, 0)); x = np.empty((2 * M, 0))
    while X.shape[1] < N:
        Xn = 400 * rng.random((3, 2 * N)) - 200
        xh = [P @ np.vstack([Xn, np.ones(2 * N)]) for P in Ps]
        xn = np.vstack([h[0:2] / h[2:3] for h in xh]) + noise * rng.standard_normal((2 * M, 2 * N))
        inside = np.all((xn[0::2] <= 36 * pix) & (xn[0::2] >= 0) & (xn[1::2] <= 24 * pix) & (xn[1::2] >= 0), axis=0)
        X = np.hstack([X, Xn[:, inside]]); x = np.hstack([x, xn[:, inside]])
    R_t = np.vstack([R @ np.hstack([Rs[0].T, (Cs[0] - c).reshape(3, 1)]) for R, c in zip(Rs, Cs)])
    R_t[0:3] = np.eye(3, 4)
    return x[:, :N].copy(), np.vstack([K] * M), R_t, (Rs[0] @ (X[:, :N] - Cs[0].reshape(3, 1)))

"""Comparison rules shared by the parity tests: T up to global
sign; R, t, Reconst directly; relative tolerance stated at each call site."""
import numpy as np


def rel_err_T(T, T_ref):
    """max |s*T - T_ref| / max|T_ref| with s the global sign aligning T to T_ref."""
    T = np.asarray(T); T_ref = np.asarray(T_ref)
    s = np.sign(np.sum(T * T_ref))
    return float(np.max(np.abs(s * T - T_ref)) / np.max(np.abs(T_ref)))


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def golden_cases(npz, prefix="c"):
    i = 0
    while "%s%d_meta" % (prefix, i) in npz:
        yield i, "%s%d_" % (prefix, i)
        i += 1


# ---- Pi methods: the sign/basis conventions of the HIP kernel, as hooks for the oracle --------------------
def _cross4(r0, r1, r2):
    Mx = np.stack([r0, r1, r2])
    n = np.array([(-1) ** j * np.linalg.det(np.delete(Mx, j, axis=1)) for j in range(4)])
    return n / np.linalg.norm(n)


def kernel_null_convention(A):
    """null(A) with the conventions of csrc/pi_kernel.h: generalised cross product for a 3x4 (or 3 row vectors);
    for the 2x4 of PiColPoseEstimation.m:63 the projection of the coordinate axis farthest from the plane and its
    orthogonal complement.  MATLAB leaves these signs / this basis to svd."""
    from oracle import tft_oracle as O
    if A.shape == (3, 4):
        return _cross4(A[0], A[1], A[2]).reshape(4, 1)
    if A.shape == (2, 4):
        n1, n2 = A[0], A[1]
        q2 = n2 - (n1 @ n2) * n1
        q2 = q2 / np.linalg.norm(q2)
        best, v3 = -1.0, None
        for ax in range(4):
            e = np.eye(4)[ax] - n1[ax] * n1 - q2[ax] * q2
            if e @ e > best:
                best, v3 = e @ e, e / np.linalg.norm(e)
        return np.stack([v3, _cross4(n1, q2, v3)], axis=1)
    return O.null(A)


def oracle_in_kernel_convention(method, Cb, CalM, init_p, init_x, tol=1e-8):
    """Runs the oracle's Pi / PiCol method under the svd sign convention that reproduces the kernel's start
    (init_p, init_x): the kernel's null-space construction and one of the four sign choices of the projective
    cameras P2, P3 of linearTFT.  Returns (outputs, max deviation of the start) -- the start must match to `tol`
    up to the sign of each pi vector (an overall sign of P_k flips all pi vectors of view k)."""
    from oracle import tft_oracle as O
    fn = getattr(O, method)
    best = (np.inf, None)
    for s2 in (1.0, -1.0):
        for s3 in (1.0, -1.0):
            p0, xe = fn(Cb, CalM, null=kernel_null_convention, cam_signs=(s2, s3), init_only=True)
            d = 0.0
            for blk in range(9):
                a, o = init_p[3 * blk:3 * blk + 3], p0[3 * blk:3 * blk + 3]
                d = max(d, min(np.abs(a - o).max(), np.abs(a + o).max()))
            d = max(d, np.abs(init_x - xe).max())
            if d < best[0]:
                best = (d, (s2, s3))
    assert best[0] < tol, "no sign convention reproduces the kernel's start (best deviation %.2e)" % best[0]
    return fn(Cb, CalM, True, null=kernel_null_convention, cam_signs=best[1]), best[0]


# ---- recover_R_t: conventions of svd(E) ------------------------------------------------------------------------
def vote_patterns(v):
    """The four orders in which the candidate scores [a,-a,-b,b] of recover_R_t (R_t_from_TFT.m:92-104) can come out,
    depending on the signs svd(E) gives U(:,3) and V(:,3) (E has rank 2, both are free): flipping U(:,3) swaps R <-> Rp
    and negates t, flipping V(:,3) swaps R <-> Rp."""
    a, b = v[0], v[3]
    return [[a, -a, -b, b], [-b, b, a, -a], [b, -b, -a, a], [-a, a, b, -b]]


def votes_match(kernel_votes, oracle_votes):
    k = [float(x) for x in kernel_votes]
    return any(k == [float(x) for x in p] for p in vote_patterns(list(oracle_votes)))


def has_vote_tie(v):
    """True when the `>=` rule of recover_R_t has to break a tie between the two rotations (|a| == |b|): the winner then
    depends on the sign convention of svd(E), which MATLAB does not specify."""
    return abs(v[0]) == abs(v[3])


def oracle_under_conventions(fn, Cb, CalM):
    """Outputs of an oracle pose method under the 16 sign conventions of its two svd(E) calls; the default convention
    (numpy's LAPACK as is) comes first."""
    from oracle import tft_oracle as O
    outs = []
    try:
        for s2 in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
            for s3 in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
                O.set_E_svd_signs(None if (s2 == (1, 1) and s3 == (1, 1)) else [s2, s3])
                try:
                    outs.append(fn(Cb, CalM))
                except Exception:
                    outs.append(None)
    finally:
        O.set_E_svd_signs(None)
    return outs


def pose_err(out_b, ref):
    """max relative deviation of (T up to sign, R_t_2, R_t_3) from an oracle result tuple (R_t_2, R_t_3, Reconst, T, ...)."""
    return max(rel_err_T(out_b["T"], ref[3]), rel_err(out_b["R_t_2"], ref[0]), rel_err(out_b["R_t_3"], ref[1]))


def pose_err_any_convention(out_b, fn, Cb, CalM):
    """(error against the default convention, error against the best of the 16 conventions).  The second is what a
    triplet with a cheirality-vote tie is held to: its reference result is not unique."""
    refs = oracle_under_conventions(fn, Cb, CalM)
    e0 = pose_err(out_b, refs[0]) if refs[0] is not None else float("inf")
    eb = min([pose_err(out_b, r) for r in refs if r is not None] or [float("inf")])
    return e0, eb


EPIPOLE_CONVENTIONS = [(a, b, c) for c in (1, -1) for a in (1, -1) for b in (1, -1)]      # (1, 1, 1) first: numpy's LAPACK as is


def oracle_under_epipole_conventions(fn, Cb, CalM):
    """Outputs of an oracle pose method under the eight sign conventions of linearTFT's three singular vectors that feed the cameras
    (`V(:,3)` of the two epipole svd calls, linearTFT.m:71-79, and `V(:,end)` of the constrained solve, :84; MATLAB leaves each
    sign open).  They flip e21 with a(10:18), e31 with a(1:9), and T with a(1:18): always a valid camera pair of the same tensor.
    Only Nordberg's parameterisation feels them: its rotations are built from those cameras through orth(), a NONLINEAR function
    of the convention (U, V, W each change by a half-turn), so Gauss-Helmert iterates differ at second order in the step (1e-5 ..
    4e-4 when the loop stops after one update) although the fixed point is the same."""
    from oracle import tft_oracle as O
    outs = []
    try:
        for sg in EPIPOLE_CONVENTIONS:
            O.set_epipole_signs(None if sg == (1, 1, 1) else sg)
            outs.append(fn(Cb, CalM))
    finally:
        O.set_epipole_signs(None)
    return outs

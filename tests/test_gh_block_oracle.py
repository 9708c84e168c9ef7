"""The block-structured Gauss-Helmert restatement (the algebra of the HIP kernel) against the dense
restatement of Gauss_Helmert.m: agreement is statistical (see the module docstrings)."""
import os

import numpy as np

from oracle import gh_block_oracle as G, tft_oracle as O
from helpers import rel_err_T, rel_err


def test_ressl_model_matches_dense_callback(golden_dir):
    g = np.load(os.path.join(golden_dir, "synthetic_gh.npz"))
    C = g["c0_Corresp"][0].T.copy()
    s = G.ressl_setup(C)
    f, gg, A, B, Cc, _ = O._ressl_constraintsGH(s["x_est"], s["p0"], s["Ind"])
    T, D, gm, Cm = G.ressl_model(s["p0"], s["Ind"])
    for i in range(C.shape[1]):
        fi, Bi, Ki, hi = G.point_blocks(s["x_est"][6 * i:6 * i + 6], T)
        Ap = np.kron(hi.reshape(1, 3), Ki.T)
        assert np.allclose(fi, f[4 * i:4 * i + 4], atol=1e-13)
        assert np.allclose(Bi, B[4 * i:4 * i + 4, 6 * i:6 * i + 6], atol=1e-13)
        assert np.allclose(Ap @ D, A[4 * i:4 * i + 4], atol=1e-12)
        assert np.linalg.matrix_rank(Bi @ Bi.T, tol=1e-9) == 3            # a point triplet carries 3 constraints
    assert np.allclose(Cm, Cc) and np.allclose(gm, gg)


def test_block_gh_tracks_dense_gh(golden_dir):
    g = np.load(os.path.join(golden_dir, "synthetic_gh.npz"))
    for pre, tol in (("c0_", 1e-2), ("c2_", 2e-3)):
        C, CalM = g[pre + "Corresp"], g[pre + "CalM"]
        for b in range(C.shape[0]):
            R2, R3, Rec, T, it = G.ResslTFTPoseEstimation_blocks(C[b].T.copy(), CalM)
            assert abs(it - int(g[pre + "ressl_iter"][b])) <= 2
            assert rel_err_T(T, g[pre + "ressl_T"][b]) < tol and rel_err(R3, g[pre + "ressl_Rt3"][b]) < tol

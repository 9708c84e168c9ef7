"""Export the INPUTS of the committed parity fixtures as one MATLAB file, for matlab/make_reference_golden.m.

    python tools/export_reference_inputs.py            # -> matlab/reference_pin/reference_inputs.mat

The file holds inputs only -- correspondences, calibration, which of the reference's methods a case is meant for -- taken from
tests/golden/{synthetic_linear,synthetic_gh,optimf,pi,epfl}.npz.  It contains nothing of the reference and nothing computed by this
repo's oracle or kernels: the expected values come from the reference itself, run by someone who has MATLAB."""
import os
import sys

import numpy as np
from scipy.io import savemat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
TFT = ["LinearTFTPoseEstimation", "ResslTFTPoseEstimation", "NordbergTFTPoseEstimation", "FaugPapaTFTPoseEstimation", "PiPoseEstimation",
       "PiColPoseEstimation"]
FM = ["LinearFPoseEstimation", "OptimFPoseEstimation"]
MAX_PER_CASE = 2          # triplets taken from every fixture case


def cases():
    out = []

    def add(name, C, CalM, methods):
        C = np.asarray(C, dtype=np.float64)
        if C.shape[0] != 6:
            C = C.T                                     # fixtures store (N, 6); the reference takes 6 x N
        out.append(dict(name=name, Corresp=np.ascontiguousarray(C), CalM=np.asarray(CalM, dtype=np.float64), methods=np.array(methods, dtype=object)))

    g = np.load(os.path.join(G, "synthetic_linear.npz"))
    i = 0
    while "c%d_meta" % i in g:
        C = g["c%d_Corresp" % i]; N = C.shape[1]
        for b in range(min(MAX_PER_CASE, C.shape[0])):
            add("synthetic_linear/c%d/b%d" % (i, b), C[b], g["c%d_CalM" % i], ["LinearTFTPoseEstimation"] + (["LinearFPoseEstimation", "OptimFPoseEstimation"] if N >= 8 else []))
        i += 1
    g = np.load(os.path.join(G, "synthetic_gh.npz"))
    i = 0
    while "c%d_meta" % i in g:
        C = g["c%d_Corresp" % i]
        for b in range(min(MAX_PER_CASE, C.shape[0])):
            add("synthetic_gh/c%d/b%d" % (i, b), C[b], g["c%d_CalM" % i], TFT[:5] + FM)
        i += 1
    g = np.load(os.path.join(G, "pi.npz"))
    i = 0
    while "p%d_meta" % i in g:
        C = g["p%d_Corresp" % i]
        for b in range(min(MAX_PER_CASE, C.shape[0])):
            add("pi/p%d/b%d" % (i, b), C[b], g["p%d_CalM" % i], ["PiPoseEstimation", "PiColPoseEstimation", "LinearTFTPoseEstimation"])
        i += 1
    g = np.load(os.path.join(G, "epfl.npz"), allow_pickle=True)
    i = 0
    while "t%d_name" % i in g:
        add("epfl/t%d/%s" % (i, str(g["t%d_name" % i])), g["t%d_sample" % i], g["t%d_CalM" % i], TFT[:5] + FM)
        i += 1
    return out


def main():
    cs = cases()
    dst = os.path.join(ROOT, "matlab", "reference_pin", "reference_inputs.mat")
    arr = np.empty((len(cs),), dtype=[("name", object), ("Corresp", object), ("CalM", object), ("methods", object)])
    for k, c in enumerate(cs):
        arr[k] = (c["name"], c["Corresp"], c["CalM"], c["methods"])
    savemat(dst, {"cases": arr, "format_version": 1.0}, do_compression=True)
    print("%d cases -> %s (%.1f KB)" % (len(cs), dst, os.path.getsize(dst) / 1024.0))


if __name__ == "__main__":
    sys.exit(main())

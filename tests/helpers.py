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

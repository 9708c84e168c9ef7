"""
The reference-held pin ("parity unpinned" until such a file exists).

tests/golden/reference_golden.mat is produced by someone who has MATLAB (or Octave) and a checkout of LauraFJulia/TFT_vs_Fund:

    python tools/export_reference_inputs.py                    # inputs of the committed fixtures -> matlab/reference_pin/reference_inputs.mat
    matlab -batch "cd matlab/reference_pin; make_reference_golden('/path/to/TFT_vs_Fund')"
    cp matlab/reference_pin/reference_golden.mat tests/golden/

It holds what the REFERENCE's own .m files return on those inputs (R_t_2, R_t_3, Reconst, T, iter, tic/toc seconds) for the eight pose
methods.  When the file is present the numpy oracle (CPU suite) and the HIP kernels (-m gpu) are compared with it; when it is absent --
it cannot be made in the build container, which has neither MATLAB nor Octave -- these tests skip and parity stays "partial".
The reader and the comparison rules themselves are exercised on every run against a file of the same layout written from the oracle.
"""
import os

import numpy as np
import pytest

from helpers import rel_err_T, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_golden.mat")
LINEAR = ("LinearTFTPoseEstimation", "LinearFPoseEstimation")
# north_star: 1e-6 relative on T / F entries and recovered R, t for the linear paths.  The Gauss-Helmert methods amplify MATLAB's own
# rounding of pinv(W + 1e-12 I): their gate is the spread the LAPACK oracle shows against the 50-digit iteration
TOL = {"LinearTFTPoseEstimation": 1e-6, "LinearFPoseEstimation": 1e-6, "OptimFPoseEstimation": 1e-5}
TOL_GH = 2e-3


def load_reference(path=GOLDEN):
    """-> (list of cases, info).  A case: dict(name, N, methods={method: dict(R_t_2, R_t_3, Reconst, T, iter, seconds, err)})."""
    from scipy.io import loadmat
    m = loadmat(path, squeeze_me=True, struct_as_record=False)
    res = np.atleast_1d(m["results"])
    cases = []
    for r in res:
        methods = {}
        for f in r._fieldnames:
            if f in ("name", "N"):
                continue
            o = getattr(r, f)
            methods[f] = dict(R_t_2=np.asarray(o.R_t_2, dtype=float), R_t_3=np.asarray(o.R_t_3, dtype=float), Reconst=np.asarray(o.Reconst, dtype=float),
                              T=np.asarray(o.T, dtype=float), iter=float(o.iter), seconds=float(o.seconds), err=str(o.err) if np.size(o.err) else "")
        cases.append(dict(name=str(r.name), N=int(r.N), methods=methods))
    info = m.get("info")
    return cases, info


def load_inputs():
    from scipy.io import loadmat
    path = os.path.join(ROOT, "matlab", "reference_pin", "reference_inputs.mat")
    m = loadmat(path, squeeze_me=True, struct_as_record=False)
    return {str(c.name): (np.asarray(c.Corresp, dtype=float), np.asarray(c.CalM, dtype=float)) for c in np.atleast_1d(m["cases"])}


def compare(candidate, cases, inputs):
    """candidate(method, Corresp 6xN, CalM) -> (R_t_2, R_t_3, Reconst, T, iter) or None (method not applicable).
    Returns [(case, method, deviation, iter_candidate, iter_reference)]; asserts nothing."""
    rows = []
    for c in cases:
        C, CalM = inputs[c["name"]]
        for method, ref in c["methods"].items():
            if ref["err"]:
                continue
            got = candidate(method, C, CalM)
            if got is None:
                continue
            R2, R3, Rec, T, it = got
            dev = max(rel_err_T(T, ref["T"]), rel_err(R2, ref["R_t_2"]), rel_err(R3, ref["R_t_3"]))
            if ref["Reconst"].size:
                dev = max(dev, rel_err(Rec, ref["Reconst"]))
            rows.append((c["name"], method, dev, float(it), ref["iter"]))
    return rows


def check(rows):
    assert rows, "nothing compared"
    worst = {}
    for name, method, dev, it, it_ref in rows:
        worst[method] = max(worst.get(method, 0.0), dev)
        if method in LINEAR:
            assert it == 0 and it_ref == 0
    for method, dev in worst.items():
        assert dev < TOL.get(method, TOL_GH), (method, dev, [r for r in rows if r[1] == method and r[2] == dev])
    return worst


def _oracle_candidate(method, C, CalM):
    from oracle import tft_oracle as O
    f = getattr(O, method, None)
    if f is None:
        return None
    return f(C.copy(), CalM)


def _write_file_in_the_reference_layout(path, names, inputs, produce):
    """A file with the layout make_reference_golden.m saves (cell array of structs, one field per method), filled by `produce`."""
    from scipy.io import savemat
    res = np.empty((len(names),), dtype=object)
    for k, name in enumerate(names):
        C, CalM = inputs[name]
        r = {"name": name, "N": float(C.shape[1])}
        for method in ("LinearTFTPoseEstimation", "LinearFPoseEstimation"):
            if method == "LinearFPoseEstimation" and C.shape[1] < 8:
                continue
            R2, R3, Rec, T, it = produce(method, C, CalM)
            r[method] = {"R_t_2": R2, "R_t_3": R3, "Reconst": Rec, "T": T, "iter": float(it), "seconds": 0.5, "err": ""}
        res[k] = r
    savemat(path, {"results": res, "info": {"release": "synthetic (written by the test from the oracle)", "threads": 1.0}, "format_version": 1.0})


def test_reader_and_comparison_rules_on_a_file_of_the_same_layout(tmp_path):
    """The machinery a real reference_golden.mat will go through, end to end, on a stand-in written from the oracle: the reader, the
    pairing of cases with the exported inputs, the sign rule for T, the tolerances -- and that a wrong value is caught."""
    inputs = load_inputs()
    names = [n for n in inputs if n.startswith("synthetic_linear/")][:6]
    path = str(tmp_path / "reference_golden.mat")
    _write_file_in_the_reference_layout(path, names, inputs, _oracle_candidate)
    cases, info = load_reference(path)
    assert [c["name"] for c in cases] == names and "LinearTFTPoseEstimation" in cases[0]["methods"]
    worst = check(compare(_oracle_candidate, cases, inputs))
    assert max(worst.values()) < 1e-12

    def off_by_a_bit(method, C, CalM):
        R2, R3, Rec, T, it = _oracle_candidate(method, C, CalM)
        return R2, R3 * (1 + 1e-5), Rec, -T, it                              # global sign of T is free; a 1e-5 change of R_t_3 is not
    with pytest.raises(AssertionError):
        check(compare(off_by_a_bit, cases, inputs))


def test_exported_inputs_are_the_committed_fixtures():
    """matlab/reference_pin/reference_inputs.mat (committed) is what tools/export_reference_inputs.py writes from tests/golden/*.npz today."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_reference_inputs as E
    inputs = load_inputs()
    cs = E.cases()
    assert len(cs) == len(inputs) >= 40
    for c in cs:
        C, CalM = inputs[c["name"]]
        assert np.array_equal(C, c["Corresp"]) and np.array_equal(CalM, c["CalM"]) and C.shape[0] == 6


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/reference_golden.mat absent: run matlab/reference_pin/make_reference_golden.m "
                                                       "with MATLAB/Octave and a checkout of the reference (INTEGRATION.md 6)")
def test_oracle_against_the_reference():
    cases, _ = load_reference()
    worst = check(compare(_oracle_candidate, cases, load_inputs()))
    print("oracle vs reference, worst deviation per method:", worst)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/reference_golden.mat absent (see test_oracle_against_the_reference)")
def test_hip_path_against_the_reference(gpu_ctx):
    def hip(method, C, CalM):
        out = gpu_ctx.pose_batch(method, np.ascontiguousarray(C.T)[None], CalM, reconst=True)
        if out["status"][0] != 0:
            return None
        return out["R_t_2"][0], out["R_t_3"][0], out["Reconst"][0], out["T"][0], out["iter"][0]
    cases, _ = load_reference()
    worst = check(compare(hip, cases, load_inputs()))
    print("HIP path vs reference, worst deviation per method:", worst)

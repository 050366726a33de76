"""FEAST with HipVector on the GPU: the reference's test problem (unittests/test_feast.py) and
Polizzi's Fortran known-answer data (unittests/test_feast_fortran.py).  Complex contour solves
run as device-resident GCROT(m,k) on (re, im) pairs; the reference's exact "pardiso" branch is
replaced by GCROT converged to 1e-13 on the 4 x 4 known-answer system."""
import math

import numpy as np
import pytest

from conftest import load_golden
from eigensolvers_amd import feast as pf
from test_feast_cpu import ORDER, read_fortran

pytestmark = pytest.mark.gpu


def _opts(tol, atol=None, it=1000):
    d = {"linearSolver": "gcrotmk", "linearIter": it, "linear_tol": tol}
    if atol is not None:
        d["linear_atol"] = atol
    return {"linearSystemArgs": d}


def test_complex_shift_solve_and_fortran_known_answers(hip):
    A, Y1 = read_fortran()[:2]
    H = hip.HipCsrOperator.from_dense(A)
    guess = [hip.HipVector(Y1[i, :].real.copy(), _opts(1e-13, 1e-15)) for i in range(3)]
    gk, wk = pf.quadraturePointsWeights(8, "legendre", positiveHalf=False)
    theta = (-(np.pi * 0.5) * (gk - 1))[ORDER]
    wk = wk[ORDER]
    Q = [None] * 3
    for k in range(8):
        fQe, fQ = read_fortran(k)[6:8]
        z = 4.0 + math.cos(theta[k]) + 0.3j * math.sin(theta[k])
        for i in range(3):
            Qe = hip.HipVector.solve(H, guess[i], z, opType="gen")
            assert isinstance(Qe, hip.hip_vector.HipComplexVector)
            np.testing.assert_allclose(Qe.array, fQe[i], rtol=1e-5)                 # test_Qe
            Q = pf.updateQ(Q, i, pf.calculateQuadrature(H, guess[i], z, 1.0, theta[k], wk[k], 0.3), k)
        for i in range(3):
            np.testing.assert_allclose(Q[i].array, fQ[i], rtol=1e-5)                # test_Q
    # complex scalar times vector, real part, conjugate
    v = np.arange(1.0, 5.0)
    c = (0.5 - 2.0j) * hip.HipVector(v)
    np.testing.assert_allclose(c.array, (0.5 - 2.0j) * v, rtol=1e-15)
    np.testing.assert_allclose(hip.HipVector.real((1.5 + 1j) * c).array, ((1.5 + 1j) * (0.5 - 2.0j) * v).real, rtol=1e-15)
    np.testing.assert_allclose(c.conjugate().array, np.conj(c.array))
    with pytest.raises(NotImplementedError):
        hip.HipVector.solve(H, hip.HipVector(v, {"linearSystemArgs": {"linearSolver": "minres"}}), 1.0 + 1.0j)


def test_feast_reference_problem_on_device(hip):
    g = load_golden("feast_n100.npz")
    A = g["A"]
    H = hip.HipCsrOperator.from_dense(A)
    Y = [hip.HipVector(g["guess"][:, i].copy(), _opts(1e-2)) for i in range(6)]
    ev, Yf, st = pf.feastDiagonalization(H, Y, 8, "legendre", 160.0, 166.0, 1e-10, 20, writeOut=False)
    assert isinstance(ev, np.ndarray) and isinstance(Yf, list) and isinstance(Yf[0], hip.HipVector)
    assert len(Yf) == int(g["nvec"]) and abs(st["outerIter"] - int(g["outerIter"])) <= 2
    exact = np.linalg.eigvalsh(A)
    inside = pf.select_within_range(exact, 160.0, 166.0)[0]
    found = pf.select_within_range(ev, 160.0, 166.0)[0]
    assert len(inside) == len(found) == 3
    np.testing.assert_allclose(found, inside, atol=1e-4)                          # test_feast.py bound
    np.testing.assert_allclose(found, pf.select_within_range(g["ev"], 160.0, 166.0)[0], rtol=1e-6)
    S = hip.HipVector.overlapMatrix(Yf)
    np.testing.assert_allclose(S, np.eye(len(Yf)), atol=1e-5)
    # eigenvectors of the in-window states (test_feast.py::test_eigenvector, rtol 1e-2)
    w, V = np.linalg.eigh(A)
    for e in inside:
        vec = Yf[hip.find_nearest(ev, e)[0]].array
        assert abs(abs(np.vdot(V[:, hip.find_nearest(w, e)[0]], vec)) - 1) < 1e-2

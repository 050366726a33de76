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
    # test_feast.py "back-transform" and "transformationMatrix" sub-tests (:73-104): Loewdin + Ritz of the returned basis
    # give a transformation X with X^H S X = 1, and the transformed vectors are the returned ones up to a sign
    from eigensolvers_amd.subspace import basisTransformation, loewdin_transform, ritz_pairs
    Hm = hip.HipVector.matrixRepresentation(H, Yf)
    uS = loewdin_transform(S)[1]
    uSH = uS @ ritz_pairs(uS, Hm)[1]
    np.testing.assert_allclose(uSH.T.conj() @ S @ uSH, np.eye(uSH.shape[1]), atol=1e-5)
    bases = basisTransformation(Yf, uSH)
    for m in range(len(Yf)):
        ov = bases[m].vdot(Yf[m], True)
        assert abs(abs(ov) - 1) < 1e-5
        np.testing.assert_allclose(Yf[m].array, ov * bases[m].array, atol=1e-5)
    # eigenvectors of the in-window states (test_feast.py::test_eigenvector, rtol 1e-2)
    w, V = np.linalg.eigh(A)
    for e in inside:
        vec = Yf[hip.find_nearest(ev, e)[0]].array
        assert abs(abs(np.vdot(V[:, hip.find_nearest(w, e)[0]], vec)) - 1) < 1e-2


def test_exact_branch_pardiso_on_the_fortran_system_and_beyond(hip):
    """``linearSolver="pardiso"`` (numpyVector.py:166-170), the branch unittests/test_feast_fortran.py:85-117 runs: an exact
    solve of (z I - H) x = b.  On the device it is Gaussian elimination with partial pivoting in one workgroup (n <= 96).
    The Fortran known answers to the reference's rtol 1e-5, and dense systems up to the size limit against numpy.linalg."""
    A, Y1 = read_fortran()[:2]
    H = hip.HipCsrOperator.from_dense(A)
    par = lambda: {"linearSystemArgs": {"linearSolver": "pardiso"}}
    guess = [hip.HipVector(Y1[i, :].real.copy(), par()) for i in range(3)]
    gk, wk = pf.quadraturePointsWeights(8, "legendre", positiveHalf=False)
    theta = (-(np.pi * 0.5) * (gk - 1))[ORDER]
    wk = wk[ORDER]
    Q = [None] * 3
    for k in range(8):
        fQe, fQ = read_fortran(k)[6:8]
        z = 4.0 + math.cos(theta[k]) + 0.3j * math.sin(theta[k])
        for i in range(3):
            Qe = hip.HipVector.solve(H, guess[i], z, opType="gen")
            assert isinstance(Qe, hip.hip_vector.HipComplexVector) and Qe.last_solve_stats["exact"]
            np.testing.assert_allclose(Qe.array, fQe[i], rtol=1e-5)                 # test_Qe
            np.testing.assert_allclose(Qe.array, np.linalg.solve(z * np.eye(4) - A, Y1[i, :].real), rtol=1e-13)
            Q = pf.updateQ(Q, i, pf.calculateQuadrature(H, guess[i], z, 1.0, theta[k], wk[k], 0.3), k)
        for i in range(3):
            np.testing.assert_allclose(Q[i].array, fQ[i], rtol=1e-5)                # test_Q
    rng = np.random.default_rng(17)
    for n in (1, 2, 37, 96):
        B = rng.standard_normal((n, n)); B = B + B.T
        Hn = hip.HipCsrOperator.from_dense(B)
        b = rng.standard_normal(n)
        xr = hip.HipVector.solve(Hn, hip.HipVector(b.copy(), par()), 0.37)                               # real shift: real result
        assert isinstance(xr, hip.HipVector) and not isinstance(xr, hip.hip_vector.HipComplexVector)
        np.testing.assert_allclose(xr.array, np.linalg.solve(0.37 * np.eye(n) - B, b), rtol=1e-9, atol=1e-12)
        xg = hip.HipVector.solve(Hn, hip.HipVector(b.copy(), par()), 0.37, reverseGF=True)               # H - sigma (numpyVector.py:168-169)
        np.testing.assert_allclose(xg.array, -xr.array, rtol=1e-12, atol=1e-14)
        z = 0.37 + 0.8j
        bc = b + 1j * rng.standard_normal(n)
        xc = hip.HipVector.solve(Hn, hip.HipVector(bc.copy(), par()), z)                                 # complex right-hand side
        np.testing.assert_allclose(xc.array, np.linalg.solve(z * np.eye(n) - B, bc), rtol=1e-10, atol=1e-13)
    with pytest.raises(NotImplementedError):                                                             # the size limit is loud
        big = hip.HipCsrOperator.from_dense(np.eye(97))
        hip.HipVector.solve(big, hip.HipVector(np.ones(97), par()), 0.5)
    with pytest.raises(np.linalg.LinAlgError):                                                           # singular: sigma is an eigenvalue
        hip.HipVector.solve(hip.HipCsrOperator.from_dense(np.diag([1.0, 2.0, 3.0])), hip.HipVector(np.ones(3), par()), 2.0)


def test_complex_solves_from_an_initial_guess(hip, gapped4000):
    """x0 on the complex paths (numpyVector.py:161,163 hand it to SciPy): complex shift + GCROT from a complex guess, and a
    complex right-hand side with a real shift + MINRES (the halves take the halves of the guess) - against SciPy."""
    import scipy.sparse.linalg as spla
    Hh = gapped4000[0]
    n = Hh.shape[0]
    H = hip.HipCsrOperator.from_scipy(Hh)
    rng = np.random.default_rng(4)
    b = rng.standard_normal(n); b /= np.linalg.norm(b)
    x0 = 0.2 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    z = 0.02 + 0.15j
    lin = spla.LinearOperator((n, n), matvec=lambda v: z * v - Hh @ v, dtype=np.complex128)
    ref, info = spla.gcrotmk(lin, b.astype(complex), x0, rtol=1e-9, atol=1e-12, maxiter=1000)
    W = hip.HipVector.solve(H, hip.HipVector(b.copy(), _opts(1e-9, 1e-12)), z, x0)
    assert info == 0 and isinstance(W, hip.hip_vector.HipComplexVector)
    assert np.linalg.norm(W.array - ref) <= 1e-7 * np.linalg.norm(ref)
    assert np.linalg.norm(z * W.array - Hh @ W.array - b) <= 2e-9
    # starting AT the solution costs (almost) nothing
    W2 = hip.HipVector.solve(H, hip.HipVector(b.copy(), _opts(1e-9, 1e-12)), z, W)
    assert W2.last_solve_stats["iterations"] <= 2
    bc = b + 1j * rng.standard_normal(n) / np.sqrt(n)
    mo = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 3000, "linear_tol": 1e-10}}
    linr = spla.LinearOperator((n, n), matvec=lambda v: 0.02 * v - Hh @ v, dtype=np.float64)
    rr = spla.minres(linr, bc.real, x0.real, rtol=1e-10, maxiter=3000)[0]
    ri = spla.minres(linr, bc.imag, x0.imag, rtol=1e-10, maxiter=3000)[0]
    Wc = hip.HipVector.solve(H, hip.HipVector(bc.copy(), mo), 0.02, hip.HipVector(x0.copy()))
    assert np.linalg.norm(Wc.array - (rr + 1j * ri)) <= 1e-8 * np.linalg.norm(rr + 1j * ri)


# ---------------------------------------------------------------- complex128 input vectors (numpyVector.py:89-93)
def test_complex_vectors_match_the_ndarray_backend(hip, gapped4000):
    """``HipVector(complex array)`` gives a complex device vector (two real halves) with the semantics of a
    complex-dtype NumpyVector: conjugated ``vdot`` / bilinear ``vdot(conjugate=False)``, complex scalars and
    coefficients, Hermitian Gram builders, the bilinear Gram-Schmidt sweep incl. its ``None`` exit, and the
    solves (real shift + MINRES: the two halves in lock step; complex shift + GCROT).  Checked against
    ``oracle.numpy_vector.RefVector`` (the restatement of NumpyVector) on the same complex data."""
    from oracle.numpy_vector import RefVector
    Hh, _ = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    rng = np.random.default_rng(23)
    n = 4000
    zs = [rng.standard_normal(n) + 0.3j * rng.standard_normal(n) for _ in range(5)]     # Re(x.x) > 0: the bilinear sweep keeps them
    opts = lambda: {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}}
    D = [hip.HipVector(z.copy(), opts()) for z in zs]
    R = [RefVector(z.copy(), opts()) for z in zs]
    assert all(type(d) is hip.HipComplexVector and d.dtype == np.complex128 and len(d) == n for d in D)
    assert issubclass(hip.HipComplexVector, hip.AbstractVector) and not hip.HipComplexVector.__abstractmethods__
    scale = float(np.sum(np.abs(zs[0] * zs[1])))
    for conj in (True, False):
        assert abs(D[0].vdot(D[1], conjugate=conj) - R[0].vdot(R[1], conjugate=conj)) <= 1e-13 * scale
    xr = hip.HipVector(zs[2].real.copy())                                # mixed: complex against a real device vector
    assert abs(D[0].vdot(xr) - np.vdot(zs[0], zs[2].real)) <= 1e-13 * scale
    assert abs(D[0].norm() - R[0].norm()) <= 1e-14 * R[0].norm()
    np.testing.assert_allclose((D[0] * (0.3 - 1.7j)).array, (R[0] * (0.3 - 1.7j)).array, rtol=1e-15, atol=1e-15)
    np.testing.assert_allclose(((2 + 1j) * D[0]).array, ((2 + 1j) * R[0]).array, rtol=1e-15, atol=1e-15)
    np.testing.assert_allclose((D[0] / (1.5 + 0.5j)).array, (R[0] / (1.5 + 0.5j)).array, rtol=4e-16, atol=1e-15)
    np.testing.assert_array_equal(D[0].real().array, zs[0].real)
    np.testing.assert_array_equal(D[0].conjugate().array, zs[0].conj())
    np.testing.assert_array_equal(D[0].copy().array, zs[0])
    with pytest.raises(NotImplementedError):
        D[0] *= 2.0
    c = D[1].copy().normalize()
    np.testing.assert_allclose(c.array, zs[1] / np.linalg.norm(zs[1]), rtol=2e-15, atol=1e-18)
    ay = D[0].applyOp(H).array
    ref = Hh @ zs[0]
    assert np.max(np.abs(ay - ref)) <= 1e-13 * np.max(np.abs(ref))
    coeffs = [0.5 + 1j, -1.25, 2.0j, 0.125 - 0.5j, -3.0]
    np.testing.assert_allclose(hip.HipComplexVector.linearCombination(D, coeffs).array,
                               RefVector.linearCombination(R, coeffs).array, rtol=0, atol=1e-13)
    np.testing.assert_allclose(hip.HipComplexVector.overlapMatrix(D), RefVector.overlapMatrix(R), rtol=0, atol=1e-10)
    np.testing.assert_allclose(hip.HipComplexVector.matrixRepresentation(H, D), RefVector.matrixRepresentation(Hh, R), rtol=0, atol=1e-10)
    np.testing.assert_allclose(hip.HipComplexVector.extendOverlapMatrix(D, hip.HipComplexVector.overlapMatrix(D[:4])),
                               RefVector.extendOverlapMatrix(R, RefVector.overlapMatrix(R[:4])), rtol=0, atol=1e-10)
    np.testing.assert_allclose(hip.HipComplexVector.extendMatrixRepresentation(H, D, hip.HipComplexVector.matrixRepresentation(H, D[:4])),
                               RefVector.extendMatrixRepresentation(Hh, R, RefVector.matrixRepresentation(Hh, R[:4])), rtol=0, atol=1e-10)
    # the reference's Gram-Schmidt sweep with its bilinear products, incl. the lindep -> None exit
    o_d = hip.HipComplexVector.orthogonalize_against_set(D[4], D[:3])
    o_r = RefVector.orthogonalize_against_set(R[4], R[:3])
    assert o_r is not None and o_d is not None
    np.testing.assert_allclose(o_d.array, o_r.array, rtol=0, atol=1e-12)
    dep = 0.3 * zs[0] - (0.2 + 1j) * zs[1]
    assert hip.HipComplexVector.orthogonalize_against_set(hip.HipVector(dep.copy(), opts()), D[:2]) is None
    assert RefVector.orthogonalize_against_set(RefVector(dep.copy(), opts()), R[:2]) is None
    # solves: (sigma - H) x = b with complex b
    b = zs[3] / np.linalg.norm(zs[3])
    exact = np.linalg.solve(0.02 * np.eye(n) - Hh.toarray(), b)
    w = hip.HipComplexVector.solve(H, hip.HipVector(b.copy(), opts()), 0.02)
    assert type(w) is hip.HipComplexVector
    assert np.linalg.norm(w.array - exact) <= 1e-7 * np.linalg.norm(exact)
    og = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 2000, "linear_tol": 1e-9, "linear_atol": 1e-12}}
    z = 0.05 + 0.3j
    wz = hip.HipComplexVector.solve(H, hip.HipVector(b.copy(), og), z)
    exact_z = np.linalg.solve(z * np.eye(n) - Hh.toarray(), b)
    assert np.linalg.norm(wz.array - exact_z) <= 1e-7 * np.linalg.norm(exact_z)
    wz_ref = RefVector.solve(Hh, RefVector(b.copy(), og), z)
    assert np.linalg.norm(wz.array - wz_ref.array) <= 1e-7 * np.linalg.norm(exact_z)


# ---------------------------------------------------------------- one sweep for both halves of a complex operand
def _ragged_csr(rng, n, ncols, dense_row=None):
    rows = []
    for i in range(n):
        if i % 7 == 0:
            k = 0                                           # empty rows
        elif i == 11:
            k = 2600
        elif dense_row is not None and i == dense_row:
            k = 5000
        else:
            k = int(rng.integers(1, 40))
        rows.append((rng.integers(0, ncols, size=k), rng.standard_normal(k)))     # unsorted, duplicates allowed
    rowptr = np.concatenate([[0], np.cumsum([len(c) for c, _ in rows])]).astype(np.int64)
    col = np.concatenate([c for c, _ in rows]).astype(np.int32)
    val = np.concatenate([v for _, v in rows])
    return rowptr, col, val


def test_pair_product_one_sweep_for_a_complex_operand(hip, monkeypatch):
    """``hipeig_spmv_shift_pair``: y = sign*(z x - H x) for complex x, z and the real operator (the GCROT matvec of
    the contour solves, feast.py:83-90 -> numpyVector.py:152-161).  On operators that take the column-window
    blocked kernel both halves share one sweep of the index / value stream (interleaved operand, two accumulators
    per row); the result must agree with the complex product on the host to 2e-14 of the row's absolute sum - for
    ragged rows (empty, longer than a batch, duplicates, unsorted), several column windows and several row units,
    a rectangular slab applied to a full-length operand, both signs and the plain product.  Small or pinned
    operators keep the two-sweep form with the same result."""
    import scipy.sparse as sp
    rng = np.random.default_rng(31)
    n = 300_000                                             # 3 column windows of 2^17; units of <= 10112 rows
    rowptr, col, val = _ragged_csr(rng, n, n, dense_row=150_001)
    A = sp.csr_matrix((val.copy(), col.copy(), rowptr.copy()), shape=(n, n))
    Aabs = sp.csr_matrix((np.abs(val), col.copy(), rowptr.copy()), shape=(n, n))
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = A @ x
    scale = Aabs @ np.abs(x) + np.abs(x) + 1e-300
    H = hip.HipCsrOperator.from_csr_arrays(rowptr, col, val, n)
    H.set_variant(4)
    ctx = hip.HipContext.default()
    xr, xi = hip.HipVector(x.real.copy()), hip.HipVector(x.imag.copy())
    yr, yi = ctx.alloc(n), ctx.alloc(n)

    def result():
        return hip.HipVector(yr).array + 1j * hip.HipVector(yi).array

    z = 0.37 - 1.9j
    for reverse, sgn in ((False, 1.0), (True, -1.0)):
        H.apply_shifted_pair(z, xr._buf, xi._buf, yr, yi, reverse=reverse)
        assert H.pair_info()["fused"] and H.pair_info()["launches"] >= 1
        _w = result()
        assert np.all(np.abs(_w - sgn * (z * x - ref)) <= 2e-14 * abs(z) * scale)
    H.apply_pair(xr._buf, xi._buf, yr, yi)
    got = result()
    assert np.all(np.abs(got - ref) <= 2e-14 * scale)
    assert np.all(got[::7] == 0.0)                          # empty rows
    # the two-sweep form (what a small / pinned / partitioned operator runs) gives the same numbers to rounding
    monkeypatch.setenv("HIPEIG_PAIR_SWEEP", "0")
    H.apply_shifted_pair(z, xr._buf, xi._buf, yr, yi)
    assert not H.pair_info()["fused"]
    assert np.all(np.abs(result() - (z * x - ref)) <= 2e-14 * abs(z) * scale)
    monkeypatch.delenv("HIPEIG_PAIR_SWEEP")
    # complex device vectors reach it through applyOp
    ay = hip.HipVector(x.copy()).applyOp(H)
    assert type(ay) is hip.HipComplexVector and H.pair_info()["fused"]
    assert np.all(np.abs(ay.array - ref) <= 2e-14 * scale)
    # rectangular slab (rows 1000..250000) applied to a full-length operand
    slab = hip.HipCsrOperator.from_scipy(A, 1000, 250_000)
    slab.set_variant(4)
    sr, si = ctx.alloc(249_000), ctx.alloc(249_000)
    slab.apply_shifted_pair(z, xr._buf, xi._buf, sr, si)
    assert slab.pair_info()["fused"]
    gs = hip.HipVector(sr).array + 1j * hip.HipVector(si).array
    assert np.all(np.abs(gs - (z * x - ref)[1000:250_000]) <= 2e-14 * abs(z) * scale[1000:250_000])
    # a small operator streams (CSR kernel): two sweeps, same contract
    m = 3000
    rp2, c2, v2 = _ragged_csr(rng, m, m)
    B = sp.csr_matrix((v2.copy(), c2.copy(), rp2.copy()), shape=(m, m))
    Hs = hip.HipCsrOperator.from_csr_arrays(rp2, c2, v2, m)
    xs = rng.standard_normal(m) + 1j * rng.standard_normal(m)
    ur, ui = ctx.alloc(m), ctx.alloc(m)
    Hs.apply_shifted_pair(z, hip.HipVector(xs.real.copy())._buf, hip.HipVector(xs.imag.copy())._buf, ur, ui)
    assert not Hs.pair_info()["fused"]
    sc2 = sp.csr_matrix((np.abs(v2), c2.copy(), rp2.copy()), shape=(m, m)) @ np.abs(xs) + np.abs(xs)
    assert np.all(np.abs(hip.HipVector(ur).array + 1j * hip.HipVector(ui).array - (z * xs - B @ xs)) <= 2e-14 * abs(z) * sc2)
    with pytest.raises(Exception):
        H.apply_shifted_pair(z, xr._buf, xi._buf, xr._buf, yi)          # in place


def test_pair_product_at_the_size_of_the_baseline_operator(hip):
    """The pair sweep on the N = 1e6 operator of BASELINE configs #2/#3/#5 (generated on the device): equal to the
    two real products to rounding, linear, and consistent with the symmetric operator (<u, H w> = <H u, w>)."""
    N = 1_000_000
    H = hip.HipCsrOperator.generate(N, 32, seed=7)
    ctx = hip.HipContext.default()
    rng = np.random.default_rng(3)
    u, w = rng.standard_normal(N), rng.standard_normal(N)
    U, W = hip.HipVector(u), hip.HipVector(w)
    Z = hip.HipVector(u + 1j * w)
    HZ = Z.applyOp(H)
    assert H.pair_info()["fused"]
    HU, HW = U.applyOp(H), W.applyOp(H)
    bound = 1e-13 * max(HU.norm(), HW.norm())
    assert hip.HipVector.linearCombination([HZ.re, HU], [1.0, -1.0]).norm() <= bound
    assert hip.HipVector.linearCombination([HZ.im, HW], [1.0, -1.0]).norm() <= bound
    assert abs(U.vdot(HZ.im) - W.vdot(HZ.re)) <= 1e-12 * U.norm() * HW.norm()          # symmetry of H
    z = 0.02 + 0.11j
    yr, yi = ctx.alloc(N), ctx.alloc(N)
    H.apply_shifted_pair(z, U._buf, W._buf, yr, yi)
    want_r = hip.HipVector.linearCombination([U, W, HU], [z.real, -z.imag, -1.0])
    want_i = hip.HipVector.linearCombination([W, U, HW], [z.real, z.imag, -1.0])
    assert hip.HipVector.linearCombination([hip.HipVector(yr), want_r], [1.0, -1.0]).norm() <= 1e-13 * want_r.norm()
    assert hip.HipVector.linearCombination([hip.HipVector(yi), want_i], [1.0, -1.0]).norm() <= 1e-13 * want_i.norm()


@pytest.mark.parametrize("N", [4000, 300_000])
def test_block_complex_shift_product_matches_the_single_products(hip, N):
    """hipeig_spmm_shift_pairs (the contour solves of one contour point share operator and shift, feast.py:198-200: four
    complex operands per pass over the operator, the shift in the block product's epilogue) against
    hipeig_spmv_shift_pair on each operand, 1..6 operands (8-wide blocks, a 4-wide remainder), both signs."""
    H = hip.HipCsrOperator.generate(N, 32, seed=7)
    rng = np.random.default_rng(N)
    ctx = hip.HipContext.default()
    z = 0.13 + 0.21j
    for npairs, reverse in ((1, False), (2, True), (3, False), (4, False), (5, True), (6, False)):
        xs = [(hip.HipVector(rng.standard_normal(N)), hip.HipVector(rng.standard_normal(N))) for _ in range(npairs)]
        ys = H.apply_shifted_pairs(z, [(a._buf, b._buf) for a, b in xs], reverse=reverse)
        for (xr, xi), (yr, yi) in zip(xs, ys):
            rr, ri = ctx.alloc(N), ctx.alloc(N)
            H.apply_shifted_pair(z, xr._buf, xi._buf, rr, ri, reverse=reverse)
            ref = hip.HipVector(rr).array + 1j * hip.HipVector(ri).array
            got = hip.HipVector(yr).array + 1j * hip.HipVector(yi).array
            assert np.max(np.abs(got - ref)) <= 1e-14 * np.max(np.abs(ref))
    if N > 100_000:
        assert H.block_info()["variant"] == "column-window-blocked"


@pytest.mark.parametrize("variant", [1, 2])
def test_eight_complex_operands_per_pass(hip, monkeypatch, variant):
    """Round 4: 8 complex operands of a contour point share one pass over the operator as a 16-wide block (128 bytes = one
    line per operand row; what config #5 runs at N = 1e7).  Forced here at a small size through its knob, in both block
    kernels (1 row-owner, 2 column-window blocked), for 5..11 operands (a 16-wide pass + remainders of every width), both
    signs, against the single pair products."""
    N = 70_001                                     # odd: the last operand row sits alone in its pair
    H = hip.HipCsrOperator.generate(N, 32, seed=7)
    H.set_block_variant(variant)
    monkeypatch.setenv("HIPEIG_PAIR_BLOCK_WIDTH", "8")
    rng = np.random.default_rng(variant)
    ctx = hip.HipContext.default()
    z = -0.07 + 0.19j
    for npairs, reverse in ((5, False), (7, True), (8, False), (11, False)):
        xs = [(hip.HipVector(rng.standard_normal(N)), hip.HipVector(rng.standard_normal(N))) for _ in range(npairs)]
        ys = H.apply_shifted_pairs(z, [(a._buf, b._buf) for a, b in xs], reverse=reverse)
        for (xr, xi), (yr, yi) in zip(xs, ys):
            rr, ri = ctx.alloc(N), ctx.alloc(N)
            H.apply_shifted_pair(z, xr._buf, xi._buf, rr, ri, reverse=reverse)
            ref = hip.HipVector(rr).array + 1j * hip.HipVector(ri).array
            got = hip.HipVector(yr).array + 1j * hip.HipVector(yi).array
            assert np.max(np.abs(got - ref)) <= 1e-14 * np.max(np.abs(ref))
    assert H.block_info()["variant"] == ("row-owner" if variant == 1 else "column-window-blocked")


def test_feast_at_the_reference_comparable_inner_tolerance(hip):
    """Config #5's recipe at N = 2e4 with gcrotmk rtol 1e-5 - the inner tolerance N = 1e7 needs (EXPERIMENTS.md R3-feast:
    looser ones do not converge there), run to the reference's stopping rule (feast.py:226-231)."""
    import warnings
    import scipy.linalg as la
    from eigensolvers_amd.generators import gapped_params
    N, m0 = 20_000, 16
    H = hip.HipCsrOperator.generate(N, 32, seed=7)
    Q = la.qr(np.random.default_rng(9).standard_normal((N, m0)), mode="economic")[0]
    o = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 4000, "linear_tol": 1e-5, "linear_atol": 1e-7,
                              "arnoldiColumnsPerPass": 4}}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = hip.feastDiagonalization(H, [hip.HipVector(Q[:, i].copy(), o) for i in range(m0)], 16, "legendre",
                                             -0.21, 0.21, 1e-4, 12, writeOut=False)
    assert st["residual"] < 1e-4 and 2 <= st["outerIter"] <= 10
    inside = np.sort(ev[(ev > -0.21) & (ev < 0.21)])
    targets = np.sort(gapped_params(N, 32, 7)["targets"])
    assert len(inside) == 16 and np.all(np.abs(inside - targets) < 2e-3)
    res = hip.true_residual_norms(H, ev, Y, m0)
    assert np.all(res < 1e-2), res


@pytest.mark.parametrize("cols", [1, 4])
def test_contour_solves_in_lock_step_equal_the_single_solves(hip, cols):
    """HipVector.solveBlock with a complex shift and gcrotmk: the right-hand sides of one contour point advance in lock
    step (gcrotmk_device_block; their products are block products), each running the unchanged complex GCROT.  Per
    right-hand side: the solution of the one-by-one solve to the solve tolerance, residual below the tolerance, about the
    same number of products (the block product rounds differently, so the count may move by a few).  cols = 4: both paths
    with the blocked Arnoldi sweep (arnoldiColumnsPerPass)."""
    N, z = 200_000, 0.02 + 0.05j
    H = hip.HipCsrOperator.generate(N, 32, seed=7)
    rng = np.random.default_rng(2)
    o = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 3000, "linear_tol": 1e-8, "linear_atol": 1e-12,
                              "arnoldiColumnsPerPass": cols}}
    bs = [hip.HipVector(rng.standard_normal(N), o) for _ in range(6)]
    for b in bs:
        b.normalize()
    one = [hip.HipVector.solve(H, b, z) for b in bs]
    its_one = [w.last_solve_stats["iterations"] for w in one]
    blk = hip.HipVector.solveBlock(H, bs, z)
    ctx = hip.HipContext.default()
    for b, w1, wb, it1 in zip(bs, one, blk, its_one):
        assert isinstance(wb, hip.hip_vector.HipComplexVector)
        a1, ab = w1.array, wb.array
        assert np.linalg.norm(ab - a1) <= 1e-6 * np.linalg.norm(a1)
        assert abs(wb.last_solve_stats["iterations"] - it1) <= max(3, it1 // 20)
        rr, ri = ctx.alloc(N), ctx.alloc(N)                       # true residual of the lock-step solution
        H.apply_shifted_pair(z, wb.re._buf, wb.im._buf, rr, ri)
        res = hip.HipVector(rr).array + 1j * hip.HipVector(ri).array - b.array
        assert np.linalg.norm(res) <= 5e-8
    # one right-hand side, blockSolve switched off, and a real shift take the one-by-one path
    assert len(hip.HipVector.solveBlock(H, bs[:1], z)) == 1

"""The oracle (oracle/) against the golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only.  This is what pins the oracle."""
import warnings

import numpy as np
import pytest
import scipy.linalg as la
import scipy.sparse.linalg as spla

from conftest import load_golden
from oracle import lanczos_ref
from oracle.minres_ref import minres as minres_ref
from oracle.numpy_vector import RefVector, SolverNotConverged
from eigensolvers_amd.generators import dense_test_matrix


def _same(actual, desired, rtol=1e-12):
    """Same container + same BLAS gives bitwise equality; the tolerance only absorbs a
    different host CPU's BLAS kernel selection (summation order of ddot/nrm2)."""
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=rtol * np.max(np.abs(np.nan_to_num(desired))),
                               equal_nan=True)


def _opts(solver, it, tol, atol=None):
    d = {"linearSolver": solver, "linearIter": it, "linear_tol": tol}
    if atol is not None:
        d["linear_atol"] = atol
    return {"linearSystemArgs": d}


def test_dense_n100_matches_reference():
    g = load_golden("lanczos_n100_seed1212.npz")
    A, _ = dense_test_matrix(100, 1212)
    e, Y, st = lanczos_ref.inexact_lanczos(A, RefVector(g["guess"].copy(), _opts("gcrotmk", 1000, 1e-4)), 30, 6, 4, 1e-6)
    _same(e, g["ev"], 1e-6)           # gcrotmk at tol 1e-4 amplifies last-bit differences to ~1e-8
    assert st["cumIter"] == int(g["cumIter"]) and st["isConverged"] == bool(g["isConverged"])
    assert st["residual"] == float(g["residual"])
    _same(Y[0].array, g["vec0"], 1e-5)


def test_block3_degenerate_matches_reference():
    g = load_golden("block3_degenerate.npz")
    A, _ = dense_test_matrix(100, 1212, g["exact"])
    Y0 = [RefVector(g["guess"][:, i].copy(), _opts("gcrotmk", 1000, 1e-4)) for i in range(3)]
    e, Y, st = lanczos_ref.inexact_lanczos(A, Y0, g["exact"][5] + 1.5, 6, 4, 1e-6)
    _same(e, g["ev"], 1e-6)
    assert st["cumIter"] == int(g["cumIter"])
    # the reference test's own assertions (unittests/test_lanczosBlock.py:54-62)
    np.testing.assert_allclose(e[:3], g["exact"][5:8], rtol=1e-6)


def test_lindep_test_case_matches_reference():
    """unittests/test_lanczosLINDEP.py:9-58 (dense n = 1200, gcrotmk rtol 1e-1, L = 100): both runs of the reference,
    through the oracle AND through the product's host driver on the oracle's vector class - Ritz values and every
    status field the reference's test reads.  With eConv = 1e-12 the run converges inside the first cycle (the
    reference's own comment: the lindep assertion 'may fail on some machines'); with 1e-18 it exhausts the cycle,
    restarts and ends with NaN Ritz values and ONE returned vector."""
    import os, sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_golden_r2 import lindep_case, LINDEP, STATUS_KEYS
    import eigensolvers_amd as ea
    ea.AbstractVector.register(RefVector)
    g = load_golden("lindep_dense_n1200.npz")
    A, y0 = lindep_case()
    for tag in ("a", "b"):
        econv = float(g[f"eConv_{tag}"])
        for run in (lambda v: lanczos_ref.inexact_lanczos(A, v, LINDEP["sigma"], LINDEP["L"], LINDEP["maxit"], econv),
                    lambda v: ea.inexactLanczosDiagonalization(A, v, LINDEP["sigma"], LINDEP["L"], LINDEP["maxit"], econv, writeOut=False)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                e, Y, st = run(RefVector(y0.copy(), _opts("gcrotmk", LINDEP["linearIter"], LINDEP["linear_tol"])))
            _same(np.asarray(e, dtype=float), g[f"ev_{tag}"], 1e-9)
            assert len(Y) == int(g[f"nvec_{tag}"])
            for k in STATUS_KEYS:
                if k == "residual":
                    assert st[k] == pytest.approx(float(g[f"residual_{tag}"]), rel=1e-3, abs=1e-15)
                else:
                    assert st[k] == g[f"{k}_{tag}"].item(), (tag, k)
    # the reference test's own assertions (:43-58): vectors returned == innerIter where lindep fired, futile restarts
    # counted only when the run stopped before maxit
    assert int(g["nvec_b"]) == int(g["innerIter_b"])


@pytest.mark.parametrize("solver", ["minres", "gcrotmk"])
def test_gapped_csr_matches_reference(gapped4000, solver):
    H, guess = gapped4000
    g = load_golden(f"gapped_csr_n4000_{solver}.npz")
    e, Y, st = lanczos_ref.inexact_lanczos(H, RefVector(guess.copy(), _opts(solver, 2000, 1e-10, 1e-12)),
                                           0.02, 8, 10, 1e-13)
    _same(e[:1], g["ev"][:1], 1e-11)
    _same(e, g["ev"], 1e-7)                # non-target Ritz values are not converged quantities
    assert st["cumIter"] == int(g["cumIter"]) and bool(st["isConverged"])
    _same(Y[0].array, g["vec0"], 1e-8)
    # and the value is right: dense eigvalsh agrees to 1e-10 relative (north-star tolerance)
    exact = np.linalg.eigvalsh(H.toarray())
    near = exact[np.argmin(abs(exact - 0.02))]
    assert abs(e[0] - near) <= 1e-10 * abs(near)


def test_block_runs_match_reference(gapped4000):
    H, _ = gapped4000
    for nb, L, maxit, tol, econv, tag in ((3, 3, 12, 1e-8, 1e-7, "block3"), (4, 3, 12, 1e-10, 1e-7, "block4_lindep")):
        g = load_golden(f"gapped_csr_n4000_{tag}.npz")
        Q = la.qr(np.random.default_rng(5).standard_normal((4000, nb)), mode="economic")[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            e, Y, st = lanczos_ref.inexact_lanczos(H, [RefVector(Q[:, i].copy(), _opts("minres", 2000, tol)) for i in range(nb)],
                                                   0.02, L, maxit, econv)
        _same(e, g["ev"], 1e-7)                              # NaNs compare equal here
        assert st["cumIter"] == int(g["cumIter"]) and st["isConverged"] == bool(g["isConverged"])
        assert len(Y) == int(g["nvec"])


def test_vector_ops_match_reference(gapped4000):
    H, guess = gapped4000
    N = 4000
    g = load_golden("spmv_n4000.npz")
    x = np.random.default_rng(11).standard_normal(N)
    _same(RefVector(x, {}).applyOp(H).array, g["y"])
    rng = np.random.default_rng(21)
    Yq = la.qr(rng.standard_normal((N, 7)), mode="economic")[0]
    qs = [RefVector(Yq[:, i].copy(), {}) for i in range(7)]
    xv = RefVector(rng.standard_normal(N), {})
    m = load_golden("mgs_step.npz")
    _same(RefVector.orthogonalize_against_set(xv, qs).array, m["out"])
    assert RefVector.orthogonalize_against_set(RefVector(Yq[:, :3] @ np.array([0.3, -0.2, 0.9]), {}), qs) is None
    vecs = [RefVector(rng.standard_normal(N), {}) for _ in range(5)]
    gm = load_golden("gram_n4000.npz")
    _same(RefVector.overlapMatrix(vecs), gm["S"])
    _same(RefVector.matrixRepresentation(H, vecs), gm["Hm"])
    _same(RefVector.extendOverlapMatrix(vecs, RefVector.overlapMatrix(vecs[:4])), gm["Sext"])
    _same(RefVector.extendMatrixRepresentation(H, vecs, RefVector.matrixRepresentation(H, vecs[:4])), gm["Hext"])
    _same(RefVector.linearCombination(vecs, [0.5, -1.25, 2.0, 0.125, -3.0]).array, gm["lincomb"])


def test_solve_matches_reference_and_raises(gapped4000):
    H, guess = gapped4000
    g = load_golden("solve_n4000_minres.npz")
    b = RefVector(guess / np.linalg.norm(guess), _opts("minres", 2000, 1e-10))
    _same(RefVector.solve(H, b, 0.02).array, g["w"])
    assert bool(g["nonconverged_raises"])
    with pytest.raises(UserWarning):
        RefVector.solve(H, RefVector(b.array.copy(), _opts("minres", 5, 1e-12)), 0.02)
    assert issubclass(SolverNotConverged, UserWarning)


def test_subspace_helpers_match_reference(gapped4000):
    H, _ = gapped4000
    g = load_golden("subspace_helpers.npz")
    rng = np.random.default_rng(21)
    la.qr(rng.standard_normal((4000, 7)), mode="economic")
    rng.standard_normal(4000)
    vecs = [RefVector(rng.standard_normal(4000), {}) for _ in range(5)]
    st, uS = lanczos_ref.lowdin_ortho_matrix(RefVector.overlapMatrix(vecs), {})
    _same(uS, g["uS"])
    ev, _ = lanczos_ref.diagonalize_hamiltonian(uS, RefVector.matrixRepresentation(H, vecs))
    _same(ev, g["evs"])
    assert lanczos_ref.eigenvalue_residual(np.array([1.0, 2.0, 3.5]), np.array([1.1, 1.9, 3.0])) == float(g["resid"])


@pytest.mark.parametrize("rtol,maxiter", [(1e-10, 2000), (1e-4, 2000), (1e-12, 7)])
def test_minres_restatement_tracks_scipy(gapped4000, rtol, maxiter):
    """oracle.minres_ref vs the SciPy routine the reference calls (numpyVector.py:163)."""
    H, guess = gapped4000
    b = guess / np.linalg.norm(guess)
    op = spla.LinearOperator(H.shape, matvec=lambda x: 0.02 * x - H @ x, dtype=np.float64)
    its = []
    xs, info_s = spla.minres(op, b, rtol=rtol, maxiter=maxiter, callback=lambda xk: its.append(1))
    xo, info_o, itn, istop = minres_ref(lambda v: 0.02 * v - H @ v, b, rtol=rtol, maxiter=maxiter)
    assert info_o == info_s
    assert itn == len(its)
    np.testing.assert_allclose(xo, xs, rtol=0, atol=1e-12 * np.linalg.norm(xs))

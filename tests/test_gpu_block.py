"""GPU parity tests of the block path (BASELINE config #3): interleaved block product in both
kernel forms, and the lock-step block MINRES against the single-vector driver and the CPU
oracle (oracle/minres_ref.py restates scipy.sparse.linalg.minres, the routine behind
NumpyVector.solve, numpyVector.py:163).

Tolerances: products 1e-14 of the row's absolute sum; per-column MINRES iteration counts and stop
codes EQUAL to the single-vector solve; iterates within max(1e-9, 100 rtol) ||x|| of the oracle
(two correctly rounded MINRES runs agree to the solve tolerance, see test_gpu_parity.py)."""
import ctypes as C
import warnings

import numpy as np
import pytest
import scipy.linalg as la
import scipy.sparse as sp

from conftest import load_golden
from eigensolvers_amd import _lib
from eigensolvers_amd.generators import gapped_csr_host
from oracle.minres_ref import minres as minres_ref

pytestmark = pytest.mark.gpu


def _within(got, ref, bound):
    got, ref, bound = np.asarray(got), np.asarray(ref), np.asarray(bound)
    err = np.abs(got - ref)
    bad = ~(err <= bound)
    assert not bad.any(), f"max excess {np.max(err - bound):.3e} at {int(np.argmax(err - bound))}, {int(bad.sum())} elements"


def _opts(it=2000, tol=1e-10, **extra):
    d = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": it, "linear_tol": tol}}
    d.update(extra)
    return d


@pytest.mark.parametrize("block_variant", [1, 2])
def test_block_product_both_kernels(hip, gapped4000, block_variant):
    """hipeig_spmm with the row-owner (1) and the column-window blocked (2) kernel, k = 1..20 operands."""
    Hh, _ = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    H.set_block_variant(block_variant)
    rng = np.random.default_rng(3)
    for k in (1, 3, 8, 11, 16, 20):
        Xh = rng.standard_normal((4000, k))
        X = [hip.HipVector(Xh[:, j].copy()) for j in range(k)]
        Y = H.apply_block([x._buf for x in X])
        ref = Hh @ Xh
        bound = 1e-14 * (np.abs(Hh) @ np.abs(Xh))
        for j in range(k):
            _within(hip.HipVector(Y[j]).array, ref[:, j], bound[:, j])
    info = H.block_info()
    assert info["variant"] == {1: "row-owner", 2: "column-window-blocked"}[block_variant]
    if block_variant == 2:
        assert info["row_blocks"] * info["rows_per_block"] >= 4000 and info["windows"] >= 1


@pytest.mark.parametrize("block_variant", [1, 2])
def test_block_product_ragged_empty_and_dense_rows(hip, block_variant, monkeypatch):
    """Empty rows, a dense row, duplicates and unsorted columns; several column windows and several
    row blocks (window and row-block sizes forced small through the tuning knobs)."""
    monkeypatch.setenv("HIPEIG_BCOO_WBITS", "9")
    monkeypatch.setenv("HIPEIG_BCOO_RW", "37")
    rng = np.random.default_rng(5)
    n = 3000
    rows = []
    for i in range(n):
        k = 0 if i % 7 == 0 else n if i == 1500 else int(rng.integers(1, 90))
        rows.append((rng.integers(0, n, size=k), rng.standard_normal(k)))
    rowptr = np.concatenate([[0], np.cumsum([len(c) for c, _ in rows])]).astype(np.int64)
    col = np.concatenate([c for c, _ in rows]).astype(np.int32)
    val = np.concatenate([v for _, v in rows])
    A = sp.csr_matrix((val.copy(), col.copy(), rowptr.copy()), shape=(n, n))
    Aabs = sp.csr_matrix((np.abs(val), col.copy(), rowptr.copy()), shape=(n, n))
    H = hip.HipCsrOperator.from_csr_arrays(rowptr, col, val, n)
    H.set_block_variant(block_variant)
    Xh = rng.standard_normal((n, 5))
    Y = H.apply_block([hip.HipVector(Xh[:, j].copy())._buf for j in range(5)])
    ref, bound = A @ Xh, 2e-14 * (Aabs @ np.abs(Xh)) + 1e-300
    for j in range(5):
        got = hip.HipVector(Y[j]).array
        _within(got, ref[:, j], bound[:, j])
        assert np.all(got[::7] == 0.0)
    if block_variant == 2:
        info = H.block_info()
        assert info["rows_per_block"] == 37 and info["windows"] == (n + 511) // 512
    # a rectangular slab applied to full-length operands (what one rank of a partition holds)
    slab = hip.HipCsrOperator.from_scipy(A, 100, 900)
    slab.set_block_variant(block_variant)
    Ys = slab.apply_block([hip.HipVector(Xh[:, j].copy())._buf for j in range(3)])
    for j in range(3):
        _within(hip.HipVector(Ys[j]).array, ref[100:900, j], bound[100:900, j])


def _block_solve(hip, H, B, sigma, it, tol, reverse=False):
    vecs = [hip.HipVector(B[:, j].copy(), _opts(it, tol)) for j in range(B.shape[1])]
    out = hip.HipVector.solveBlock(H, vecs, sigma, reverseGF=reverse)
    return out, vecs


@pytest.mark.parametrize("block_variant", [1, 2])
@pytest.mark.parametrize("k,rtol", [(8, 1e-10), (3, 1e-6), (4, 1e-8), (5, 1e-4), (11, 1e-8)])
def test_block_minres_equals_the_single_solves(hip, gapped4000, block_variant, k, rtol):
    """Column j of the lock-step solve against hipeig_minres on b_j alone and against the oracle:
    iteration count and stop code equal, iterate within the solve tolerance.  Right-hand sides of very
    different difficulty (a near-eigenvector stops after a few iterations, a zero column at once), so
    the masking of finished columns is exercised; k <= 4 runs on the 4-wide interleave, 5..8 on the 8-wide one,
    k = 11 goes through a chunk of 8 and a chunk of 3."""
    Hh, guess = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    H.set_block_variant(block_variant)
    rng = np.random.default_rng(17 + k)
    B = rng.standard_normal((4000, k))
    B /= np.linalg.norm(B, axis=0)
    w, V = np.linalg.eigh(Hh.toarray()) if k == 8 else (None, None)
    if k == 8:
        B[:, 2] = V[:, np.argmin(np.abs(w - 0.0133))] + 1e-9 * rng.standard_normal(4000)    # one MINRES step away from done
        B[:, 5] = 0.0                                                                    # beta1 = 0: x = 0, no iterations
        B[:, 6] *= 1e-7                                                                  # scaling must not matter
    X, vecs = _block_solve(hip, H, B, 0.02, 2000, rtol)
    assert len(X) == k
    for j in range(k):
        st = X[j].last_solve_stats
        if not np.any(B[:, j]):
            assert st["iterations"] == 0 and np.all(X[j].array == 0.0)
            continue
        xo, info, itn, istop = minres_ref(lambda v: 0.02 * v - Hh @ v, B[:, j], rtol=rtol, maxiter=2000)
        single = hip.HipVector.solve(H, hip.HipVector(B[:, j].copy(), _opts(2000, rtol)), 0.02)
        ss = single.last_solve_stats
        assert (st["iterations"], st["istop"]) == (ss["iterations"], ss["istop"]), f"column {j}"
        assert (st["iterations"], st["istop"]) == (itn, istop), f"column {j} vs oracle"
        xtol = max(1e-9, 100 * rtol) * np.linalg.norm(xo)
        _within(X[j].array, xo, xtol)
        _within(X[j].array, single.array, xtol)
        assert abs(st["rnorm"] - ss["rnorm"]) <= 0.05 * ss["rnorm"] + 1e-300      # the estimate moves with the summation order
        assert vecs[j].last_solve_stats is st
    if k == 8:
        its = [X[j].last_solve_stats["iterations"] for j in range(k)]
        assert its[2] < min(its[0], its[1]) and its[5] == 0               # columns really stopped at different times
    # the reverse Green's function form: (H - sigma) x = b  ->  x = -w
    Xr, _ = _block_solve(hip, H, B, 0.02, 2000, rtol, reverse=True)
    for j in range(k):
        xtol = max(1e-9, 100 * rtol) * (np.linalg.norm(X[j].array) + 1e-300)
        _within(Xr[j].array, -X[j].array, xtol)


def test_block_minres_nonconvergence_raises_and_fallbacks(hip, gapped4000):
    """numpyVector.py:175-177: a solve that hits the iteration limit raises; so does a block with such a
    column.  gcrotmk blocks and single-vector blocks take the one-by-one path."""
    Hh, guess = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    B = np.random.default_rng(2).standard_normal((4000, 4))
    vecs = [hip.HipVector(B[:, j].copy(), _opts(5, 1e-12)) for j in range(4)]
    with pytest.raises(UserWarning):
        hip.HipVector.solveBlock(H, vecs, 0.02)
    assert all(v.last_solve_stats["iterations"] == 5 and v.last_solve_stats["istop"] == 6 for v in vecs)
    for few in (1, 2):                       # below BLOCK_SOLVE_MIN right-hand sides: the one-by-one calls, bit for bit
        got = hip.HipVector.solveBlock(H, [hip.HipVector(B[:, j].copy(), _opts(2000, 1e-8)) for j in range(few)], 0.02)
        for j in range(few):
            ref = hip.HipVector.solve(H, hip.HipVector(B[:, j].copy(), _opts(2000, 1e-8)), 0.02)
            np.testing.assert_array_equal(got[j].array, ref.array)
    og = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 2000, "linear_tol": 1e-8, "linear_atol": 1e-10}}
    gs = hip.HipVector.solveBlock(H, [hip.HipVector(B[:, j].copy(), og) for j in range(2)], 0.02)
    for j in range(2):
        r = B[:, j] - (0.02 * gs[j].array - Hh @ gs[j].array)
        assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(B[:, j])


def test_block_lanczos_block8_matches_reference_golden(hip, gapped4000):
    """BASELINE config #3 at the size the reference finishes in seconds: block of 8 on the gapped CSR
    (N = 4000), the reference's own run stored by tests/golden/make_golden_r2.py.  Same eigenvalues
    (1e-10 relative where the reference converged them that far), same cumulative iteration count,
    same exit - through the lock-step solves and, for comparison, through the one-by-one solves."""
    Hh, _ = gapped4000
    g = load_golden("gapped_csr_n4000_block8.npz")
    H = hip.HipCsrOperator.from_scipy(Hh)
    Q = la.qr(np.random.default_rng(5).standard_normal((4000, 8)), mode="economic")[0]
    for block_solve in (True, False):
        v0 = [hip.HipVector(Q[:, i].copy(), _opts(2000, float(g["linear_tol"]), orthogonalization="mgs", blockSolve=block_solve))
              for i in range(8)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev, Y, st = hip.inexactLanczosDiagonalization(H, v0, 0.02, int(g["L"]), int(g["maxit"]), float(g["eConv"]),
                                                         writeOut=False)
        assert st["cumIter"] == int(g["cumIter"]) and bool(st["isConverged"]) == bool(g["isConverged"])
        assert len(Y) == int(g["nvec"])
        np.testing.assert_allclose(np.sort(ev[:8]), np.sort(g["ev"][:8]), rtol=float(g["ev_rtol"]), atol=0)
        if st["isConverged"]:
            res = hip.true_residual_norms(H, ev, Y, 8)
            assert np.all(res < 1e-3)            # eConv = 1e-8 on the eigenvalue ~ residual norm^2
            S = hip.HipVector.overlapMatrix(Y[:8])
            np.testing.assert_allclose(S, np.eye(8), atol=1e-7)

"""GPU parity tests: every HIP entry point, called through the C ABI (ctypes binding /
HipVector), against the CPU oracle and the golden vectors of the real reference.

Tolerances (fp64 path, stated per test): element-wise kernels 1e-15..1e-14 relative;
reductions 1e-13 relative to sum|a_i b_i|; inner solves 1e-9 relative to ||x||; Ritz values
1e-10 relative (the north-star bound); iteration counts equal."""
import warnings

import numpy as np
import pytest
import scipy.linalg as la
import scipy.sparse as sp

from conftest import load_golden
from eigensolvers_amd.generators import dense_test_matrix, gapped_csr_host
from oracle import lanczos_ref
from oracle.minres_ref import minres as minres_ref
from oracle.numpy_vector import RefVector

pytestmark = pytest.mark.gpu


def _within(got, ref, bound):
    """|got - ref| <= bound element-wise (bound may be an array: per-element error budget)."""
    got, ref, bound = np.asarray(got), np.asarray(ref), np.asarray(bound)
    err = np.abs(got - ref)
    bad = ~(err <= bound)
    assert not bad.any(), f"max excess {np.max(err - bound):.3e} at {int(np.argmax(err - bound))}, {int(bad.sum())} elements"


def _opts(it=2000, tol=1e-10, **extra):
    d = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": it, "linear_tol": tol}}
    d.update(extra)
    return d


# ---------------------------------------------------------------- BLAS-1 class
@pytest.mark.parametrize("n", [1, 2, 63, 64, 1000, 4097, 1 << 20, (1 << 21) + 3])
def test_blas1_against_numpy(hip, n):
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    X, Y = hip.HipVector(x), hip.HipVector(y)
    scale = np.sum(np.abs(x * y)) + 1e-300
    assert abs(X.vdot(Y) - np.dot(x, y)) <= 1e-13 * scale
    assert abs(X.vdot(Y, conjugate=False) - np.dot(x, y)) <= 1e-13 * scale
    assert abs(X.norm() - np.linalg.norm(x)) <= 1e-14 * np.linalg.norm(x)
    np.testing.assert_array_equal((X * 1.7).array, x * 1.7)              # one rounding: exact
    np.testing.assert_array_equal((-0.3 * X).array, x * -0.3)
    np.testing.assert_array_equal((X / 3.0).array, x / 3.0)              # true division: exact
    assert len(X) == n and X.dtype == np.float64 and X.maxD == 0 and X.hasExactAddition
    Z = X.copy()
    Z.normalize()
    np.testing.assert_allclose(Z.array, x / np.linalg.norm(x), rtol=4e-16, atol=0)
    np.testing.assert_array_equal(X.array, x)                            # copy() did not alias
    with pytest.raises(NotImplementedError):
        X *= 2.0
    with pytest.raises(NotImplementedError):
        X /= 2.0


@pytest.mark.parametrize("n,m", [(5, 1), (1001, 3), (4096, 7), (100003, 17), (1 << 18, 40)])
def test_tall_skinny_against_numpy(hip, n, m):
    rng = np.random.default_rng(n + m)
    Yh = rng.standard_normal((n, m))
    x = rng.standard_normal(n)
    V = [hip.HipVector(Yh[:, j].copy()) for j in range(m)]
    X = hip.HipVector(x)
    got = hip.HipVector._multi_dot(V, X)
    ref = Yh.T @ x
    _within(got, ref, 1e-13 * np.abs(Yh).T @ np.abs(x))
    c = rng.standard_normal(m)
    lc = hip.HipVector.linearCombination(V, c)
    _within(lc.array, Yh @ c, 1e-14 * (np.abs(Yh) @ np.abs(c)) + 1e-300)
    S = hip.HipVector.overlapMatrix(V)
    _within(S, Yh.T @ Yh, 1e-12 * n)
    assert np.array_equal(S, S.T)
    # block combination (basisTransformation with a matrix)
    Cm = rng.standard_normal((m, 3))
    outs = hip.basisTransformation(V, Cm)
    for k in range(3):
        _within(outs[k].array, Yh @ Cm[:, k], 1e-13 * (np.abs(Yh) @ np.abs(Cm[:, k])))


@pytest.mark.parametrize("n,ma,mb", [(777, 5, 9), (4099, 16, 16), (100001, 20, 9), (100001, 9, 20), (65536, 33, 31),
                                     (30011, 40, 70)])
def test_rectangular_gram_blocks(hip, n, ma, mb):
    """hipeig_gram with two different column sets (the <y_i, H y_j> block of matrixRepresentation):
    every accumulator-block shape of the MFMA kernel (1x1, 1x2, 2x1, 2x2, several 32-column passes)."""
    import ctypes as C
    from eigensolvers_amd import _lib
    from eigensolvers_amd.hip_vector import _ptr_table
    rng = np.random.default_rng(n + ma + 3 * mb)
    Ah, Bh = rng.standard_normal((n, ma)), rng.standard_normal((n, mb))
    A = [hip.HipVector(Ah[:, j].copy()) for j in range(ma)]
    B = [hip.HipVector(Bh[:, j].copy()) for j in range(mb)]
    ta, ka = _ptr_table([v._buf for v in A])
    tb, kb = _ptr_table([v._buf for v in B])
    M = np.empty((ma, mb))
    _lib.call("hipeig_gram", A[0].ctx.handle, n, ma, ta, mb, tb, M.ctypes.data_as(C.POINTER(C.c_double)))
    _within(M, Ah.T @ Bh, 1e-13 * (np.abs(Ah).T @ np.abs(Bh)))
    # a shared leading part: the first min(ma, mb, 16) columns of B are A's own (partly aliased tables)
    q = min(ma, mb, 16)
    mixed = A[:q] + B[q:]
    tm, km = _ptr_table([v._buf for v in mixed])
    _lib.call("hipeig_gram", A[0].ctx.handle, n, ma, ta, mb, tm, M.ctypes.data_as(C.POINTER(C.c_double)))
    Mh = np.concatenate([Ah[:, :q], Bh[:, q:]], axis=1)
    _within(M, Ah.T @ Mh, 1e-13 * (np.abs(Ah).T @ np.abs(Mh)))


@pytest.mark.parametrize("n,m,k", [(9, 2, 1), (4097, 7, 5), (100003, 33, 9), (1 << 16, 40, 17), (50001, 64, 37)])
def test_block_combination_shapes(hip, n, m, k):
    """hipeig_lincomb_block (Y*C, basisTransformation with a matrix): one pass for <= 16 outputs,
    several for more; odd lengths; up to 64 inputs."""
    rng = np.random.default_rng(n + m + k)
    Yh = rng.standard_normal((n, m))
    Cm = rng.standard_normal((m, k))
    V = [hip.HipVector(Yh[:, j].copy()) for j in range(m)]
    outs = hip.HipVector.linearCombinationBlock(V, Cm)
    ref = Yh @ Cm
    bound = 1e-14 * (np.abs(Yh) @ np.abs(Cm))
    for c in range(k):
        _within(outs[c].array, ref[:, c], bound[:, c] + 1e-300)
    for j in range(m):                                  # inputs untouched
        assert np.array_equal(V[j].array, Yh[:, j])


def test_vector_ops_match_reference_golden(hip, gapped4000):
    Hh, _ = gapped4000
    N = 4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    g = load_golden("spmv_n4000.npz")
    x = np.random.default_rng(11).standard_normal(N)
    row_scale = np.abs(Hh) @ np.abs(x)
    for variant in (1, 2, 3, 4):
        H.set_variant(variant)
        _within(hip.HipVector(x).applyOp(H).array, g["y"], 1e-14 * row_scale)
        y = hip.HipContext.default().alloc(N)
        H.apply_shifted(0.02, hip.HipVector(x)._buf, y)
        _within(hip.HipVector(y).array, g["yshift"], 1e-14 * (row_scale + 0.02 * abs(x)))
        H.apply_shifted(0.02, hip.HipVector(x)._buf, y, reverse=True)
        _within(hip.HipVector(y).array, -g["yshift"], 1e-14 * (row_scale + 0.02 * abs(x)))
    H.set_variant(0)
    rng = np.random.default_rng(21)
    Yq = la.qr(rng.standard_normal((N, 7)), mode="economic")[0]
    xv = rng.standard_normal(N)
    m = load_golden("mgs_step.npz")
    for method, tol in (("mgs", 1e-13), ("cgs2", 1e-12)):
        qs = [hip.HipVector(Yq[:, i].copy(), {"orthogonalization": method}) for i in range(7)]
        out = hip.HipVector.orthogonalize_against_set(hip.HipVector(xv.copy(), {"orthogonalization": method}), qs)
        _within(out.array, m["out"], tol)
        assert np.max(np.abs(Yq.T @ out.array)) < 1e-14 if method == "cgs2" else True
        dep = hip.HipVector(Yq[:, :3] @ np.array([0.3, -0.2, 0.9]), {"orthogonalization": method})
        assert hip.HipVector.orthogonalize_against_set(dep, qs) is None and bool(m["dep_is_none"])
    vecs = [hip.HipVector(rng.standard_normal(N)) for _ in range(5)]
    gm = load_golden("gram_n4000.npz")
    _within(hip.HipVector.overlapMatrix(vecs), gm["S"], 1e-11)
    _within(hip.HipVector.matrixRepresentation(H, vecs), gm["Hm"], 1e-11)
    np.testing.assert_allclose(hip.HipVector.extendOverlapMatrix(vecs, hip.HipVector.overlapMatrix(vecs[:4])),
                               gm["Sext"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(hip.HipVector.extendMatrixRepresentation(H, vecs, hip.HipVector.matrixRepresentation(H, vecs[:4])),
                               gm["Hext"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(hip.HipVector.linearCombination(vecs, [0.5, -1.25, 2.0, 0.125, -3.0]).array,
                               gm["lincomb"], rtol=0, atol=1e-14)


def test_block_product_matches_single_products(hip, gapped4000):
    """hipeig_spmm (interleaved tall-skinny SpMM) against k separate products, k = 1..20."""
    Hh, _ = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    rng = np.random.default_rng(3)
    for k in (1, 3, 8, 11, 16, 20):
        Xh = rng.standard_normal((4000, k))
        X = [hip.HipVector(Xh[:, j].copy()) for j in range(k)]
        Y = H.apply_block([x._buf for x in X])
        ref = Hh @ Xh
        bound = 1e-14 * (np.abs(Hh) @ np.abs(Xh))
        for j in range(k):
            _within(hip.HipVector(Y[j]).array, ref[:, j], bound[:, j])
    vecs = [hip.HipVector(rng.standard_normal(4000)) for _ in range(19)]
    M = hip.HipVector.matrixRepresentation(H, vecs)
    Vh = np.column_stack([v.array for v in vecs])
    _within(M, Vh.T @ (Hh @ Vh), 1e-11)
    assert np.array_equal(M, M.T)


# ---------------------------------------------------------------- operator edge cases
def test_spmv_ragged_empty_long_and_unsorted_rows(hip):
    rng = np.random.default_rng(5)
    n = 3000
    rows = []
    for i in range(n):
        if i % 7 == 0:
            k = 0                                   # empty rows
        elif i == 11:
            k = 2600                                # longer than one LDS tile (2048)
        elif i == 1500:
            k = n                                   # dense row
        else:
            k = int(rng.integers(1, 90))
        cols = rng.integers(0, n, size=k)           # unsorted, duplicates allowed
        rows.append((cols, rng.standard_normal(k)))
    rowptr = np.concatenate([[0], np.cumsum([len(c) for c, _ in rows])]).astype(np.int64)
    col = np.concatenate([c for c, _ in rows]).astype(np.int32)
    val = np.concatenate([v for _, v in rows])
    # copies: scipy may sum duplicates IN PLACE inside arrays it shares with the caller
    A = sp.csr_matrix((val.copy(), col.copy(), rowptr.copy()), shape=(n, n))
    x = rng.standard_normal(n)
    ref = A @ x
    scale = sp.csr_matrix((np.abs(val), col.copy(), rowptr.copy()), shape=(n, n)) @ np.abs(x) + 1e-300
    H = hip.HipCsrOperator.from_csr_arrays(rowptr, col, val, n)
    for variant in (1, 2, 3, 4):
        H.set_variant(variant)
        got = hip.HipVector(x).applyOp(H).array
        _within(got, ref, 2e-14 * scale)
        assert np.all(got[::7] == 0.0)
    # rectangular slab (rows 100..900 of the same matrix): what one rank of a partition holds
    slab = hip.HipCsrOperator.from_scipy(A, 100, 900)
    assert slab.shape == (800, n) and slab.row_offset == 100
    ys = hip.HipContext.default().alloc(800)
    for variant in (1, 2, 3, 4):                  # a slab applied to a full-length operand
        slab.set_variant(variant)
        slab.apply(hip.HipVector(x)._buf, ys)
        _within(hip.HipVector(ys).array, ref[100:900], 2e-14 * scale[100:900])
        slab.apply_shifted(0.5, hip.HipVector(x)._buf, ys)
        _within(hip.HipVector(ys).array, 0.5 * x[100:900] - ref[100:900], 2e-14 * (scale[100:900] + abs(x[100:900])))
    with pytest.raises(Exception):
        hip.HipVector.solve(slab, hip.HipVector(x[:800].copy(), _opts()), 0.02)   # not square
    # degenerate shapes
    E = hip.HipCsrOperator.from_csr_arrays(np.zeros(5, dtype=np.int64), np.zeros(0, np.int32), np.zeros(0), 4)
    np.testing.assert_array_equal(hip.HipVector(np.ones(4)).applyOp(E).array, np.zeros(4))
    with pytest.raises(Exception):
        hip.HipCsrOperator.from_csr_arrays(np.array([0, 1]), np.array([7], np.int32), np.ones(1), 4)   # col out of range
    with pytest.raises(TypeError):
        hip.HipVector(x).applyOp(A)                 # a host matrix is not a device operator


@pytest.mark.parametrize("csplit", [2, 5, 64])
def test_column_split_sweep(hip, csplit, monkeypatch):
    """TCOO-W with several workgroups per row block (column splits + combine launch; the mode picked for
    slabs of very wide operators, forced here through its tuning knob): product, fused shift and the
    MINRES iteration built on it against the CSR-stream kernel."""
    N = 300_000
    H = hip.HipCsrOperator.generate(N, 32, seed=5)
    x = hip.HipVector(np.random.default_rng(4).standard_normal(N))
    b = hip.HipVector(np.random.default_rng(5).standard_normal(N), _opts())
    b.normalize()
    H.set_variant(2)
    y_ref = x.applyOp(H)
    w_ref = hip.HipVector.solve(H, b, 0.02)
    monkeypatch.setenv("HIPEIG_TCOOW_CSPLIT", str(csplit))
    H2 = hip.HipCsrOperator.generate(N, 32, seed=5)          # the layout is built on first use, with the knob set
    H2.set_variant(4)
    y = x.applyOp(H2)
    assert H2.last_variant() == "column-window-blocked(workgroup)"
    d = hip.HipVector.linearCombination([y, y_ref], [1.0, -1.0])
    assert d.norm() <= 1e-14 * y_ref.norm()
    ctx = hip.HipContext.default()
    ys, ys_ref = ctx.alloc(N), ctx.alloc(N)
    H2.apply_shifted(0.02, x._buf, ys, reverse=True)
    H.apply_shifted(0.02, x._buf, ys_ref, reverse=True)
    d = hip.HipVector.linearCombination([hip.HipVector(ys), hip.HipVector(ys_ref)], [1.0, -1.0])
    assert d.norm() <= 1e-14 * hip.HipVector(ys_ref).norm()
    w = hip.HipVector.solve(H2, b, 0.02)
    assert w.last_solve_stats["iterations"] == w_ref.last_solve_stats["iterations"]
    d = hip.HipVector.linearCombination([w, w_ref], [1.0, -1.0])
    assert d.norm() <= 1e-8 * w_ref.norm()


def test_device_generator_is_bit_identical_to_host(hip):
    for N, nnz_row, seed in ((4000, 32, 7), (3001, 64, 11), (70001, 16, 3)):
        Hd = hip.HipCsrOperator.generate(N, nnz_row, seed=seed).to_scipy()
        Hh = gapped_csr_host(N, nnz_row, seed=seed)
        assert np.array_equal(Hd.indptr, Hh.indptr) and np.array_equal(Hd.indices, Hh.indices)
        assert np.array_equal(Hd.data, Hh.data)
    part = hip.HipCsrOperator.generate(4000, 32, seed=7, row_begin=1234, row_end=3210).to_scipy()
    assert abs(part - gapped_csr_host(4000, 32, seed=7)[1234:3210]).max() == 0.0


# ---------------------------------------------------------------- inner solve
@pytest.mark.parametrize("rtol,maxiter", [(1e-10, 2000), (1e-4, 2000), (1e-6, 2000)])
def test_minres_tracks_the_oracle(hip, gapped4000, rtol, maxiter):
    Hh, guess = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    b = guess / np.linalg.norm(guess)
    trace = []
    xo, info, itn, istop = minres_ref(lambda v: 0.02 * v - Hh @ v, b, rtol=rtol, maxiter=maxiter, trace=trace)
    B = hip.HipVector(b.copy(), _opts(maxiter, rtol))
    W = hip.HipVector.solve(H, B, 0.02)
    st = W.last_solve_stats
    assert st["iterations"] == itn and st["istop"] == istop
    # Two correctly rounded MINRES runs agree to the solve tolerance, not to rounding: the
    # Lanczos recurrence amplifies last-bit differences (summation order) as it converges, most of
    # all in the component along the eigenvector next to the shift (gap 0.0067, i.e. an error
    # amplification of ~1e3 on the residual tolerance).  Even the CPU oracle alone moves by 2e-5 ||x||
    # at rtol 1e-6 when SciPy has sorted the CSR indices of its operator in place or not.
    xtol = max(1e-9, 100 * rtol) * np.linalg.norm(xo)
    _within(W.array, xo, xtol)
    assert abs(st["rnorm"] - trace[-1]["rnorm"]) <= 0.05 * trace[-1]["rnorm"]
    assert abs(st["Anorm"] - trace[-1]["Anorm"]) <= 1e-3 * trace[-1]["Anorm"]
    # reverse Green's function: (H - sigma) x = b  ->  x = -w
    Wr = hip.HipVector.solve(H, B, 0.02, reverseGF=True)
    _within(Wr.array, -xo, xtol)
    # true residual of the returned solution: as good as the oracle's
    r = b - (0.02 * W.array - Hh @ W.array)
    ro = b - (0.02 * xo - Hh @ xo)
    assert np.linalg.norm(r) <= 1.05 * np.linalg.norm(ro) + 1e-13


def test_minres_two_and_three_kernel_forms_agree_in_every_sweep_layout(hip, monkeypatch):
    """KD riding in the epilogue of the next operator sweep (the default, two kernels per iteration) against KD as
    its own kernel (HIPEIG_MINRES_FUSE_KD=0): same expressions in the same order, so iteration count, stop code and
    the iterate agree - for every sweep layout the epilogue lives in: CSR-vector, CSR-stream, the two blocked
    layouts, the fixed-point form, and a column-split blocked sweep whose epilogue runs in the combine launch.  The
    oracle (SciPy's recurrences) is the referee for the count."""
    from eigensolvers_amd.generators import gapped_csr_host, guess_vector
    n = 300_000                                                # three column windows of the blocked layouts
    Hh = gapped_csr_host(n, 16, seed=7)
    b = guess_vector(n, 1)
    b = b / np.linalg.norm(b)
    xo, info, itn, istop = minres_ref(lambda v: 0.02 * v - Hh @ v, b, rtol=1e-8, maxiter=3000)

    def solve(variant):
        H = hip.HipCsrOperator.from_scipy(Hh)
        H.set_variant(variant)
        W = hip.HipVector.solve(H, hip.HipVector(b.copy(), _opts(3000, 1e-8)), 0.02)
        return W.array, W.last_solve_stats

    cases = [(1, None), (2, None), (3, None), (4, None), (5, None), (4, "3")]
    for variant, csplit in cases:
        if csplit:
            monkeypatch.setenv("HIPEIG_TCOOW_CSPLIT", csplit)   # read when the layout is built
        x2, s2 = solve(variant)
        monkeypatch.setenv("HIPEIG_MINRES_FUSE_KD", "0")
        x3, s3 = solve(variant)
        monkeypatch.delenv("HIPEIG_MINRES_FUSE_KD")
        if csplit:
            monkeypatch.delenv("HIPEIG_TCOOW_CSPLIT")
        assert s2["iterations"] == s3["iterations"] == itn and s2["istop"] == s3["istop"] == istop, (variant, csplit)
        scale = np.linalg.norm(xo)
        assert np.linalg.norm(x2 - x3) <= 1e-7 * scale, (variant, csplit)      # variant 4 adds a row in varying order
        assert np.linalg.norm(x2 - xo) <= 1e-6 * scale, (variant, csplit)
        if variant in (1, 2, 3, 5) and not csplit:                             # fixed summation order: bit for bit (a scalar's
            # summation tree does not depend on which kernel's prologue reduces it, common.h block_sum_partials)
            np.testing.assert_array_equal(x2, x3)


def test_minres_graph_replay_is_bit_identical(hip, gapped4000, monkeypatch):
    """HIPEIG_GRAPH=1 (read when a context is created): the captured 18-iteration chunk replayed for
    every chunk and every solve gives exactly the plain-launch result; changing the tolerance or the
    operator re-captures."""
    Hh, guess = gapped4000
    b = guess / np.linalg.norm(guess)

    def run(ctx):
        out = []
        H = hip.HipCsrOperator.from_scipy(Hh, ctx=ctx)
        for rtol in (1e-6, 1e-10, 1e-6):
            for rhs in (b, b[::-1].copy()):
                W = hip.HipVector.solve(H, hip.HipVector(rhs.copy(), _opts(2000, rtol), ctx=ctx), 0.02)
                out.append((W.last_solve_stats["iterations"], W.array))
        H2 = hip.HipCsrOperator.generate(4000, 32, seed=8, ctx=ctx)
        W = hip.HipVector.solve(H2, hip.HipVector(b.copy(), _opts(2000, 1e-8), ctx=ctx), 0.02)
        out.append((W.last_solve_stats["iterations"], W.array))
        return out

    plain = run(hip.HipContext.default())
    monkeypatch.setenv("HIPEIG_GRAPH", "1")
    graph = run(hip.HipContext(0))
    for (it_p, w_p), (it_g, w_g) in zip(plain, graph):
        assert it_p == it_g
        np.testing.assert_array_equal(w_p, w_g)


def test_solve_matches_reference_golden_and_error_behaviour(hip, gapped4000):
    Hh, guess = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    g = load_golden("solve_n4000_minres.npz")
    b = guess / np.linalg.norm(guess)
    W = hip.HipVector.solve(H, hip.HipVector(b.copy(), _opts()), 0.02)
    _within(W.array, g["w"], 1e-9 * float(g["wnorm"]))
    with pytest.raises(UserWarning):                              # numpyVector.py:175-177
        hip.HipVector.solve(H, hip.HipVector(b.copy(), _opts(5, 1e-12)), 0.02)
    with pytest.raises(NotImplementedError):
        hip.HipVector.solve(H, hip.HipVector(b.copy(), {"linearSystemArgs": {"linearSolver": "pardiso"}}), 0.02)
    with pytest.raises(Exception):
        hip.HipVector.solve(H, hip.HipVector(b.copy(), {"linearSystemArgs": {"linearSolver": "bogus"}}), 0.02)
    zero = hip.HipVector.solve(H, hip.HipVector(np.zeros(4000), _opts()), 0.02)
    assert np.all(zero.array == 0.0)                              # beta1 == 0 exit of MINRES
    opts = {}
    hip.HipVector(b.copy(), opts)
    assert opts == {}                                             # defaults go into linearSystemArgs only if given
    lsa = {"linearSystemArgs": {}}
    hip.HipVector(b.copy(), lsa)
    assert lsa["linearSystemArgs"] == {"linearSolver": "minres", "linearIter": 1000, "linear_tol": 1e-4,
                                       "linear_atol": 1e-4}      # numpyVector.py:31-36


# ---------------------------------------------------------------- the full path
def test_lanczos_single_vector_matches_reference(hip, gapped4000):
    Hh, guess = gapped4000
    g = load_golden("gapped_csr_n4000_minres.npz")
    for src in ("upload", "generate"):
        H = hip.HipCsrOperator.from_scipy(Hh) if src == "upload" else hip.HipCsrOperator.generate(4000, 32, seed=7)
        for method in ("cgs2", "mgs"):
            v0 = hip.HipVector(guess.copy(), _opts(orthogonalization=method))
            ev, Y, st = hip.inexactLanczosDiagonalization(H, v0, 0.02, 8, 10, 1e-13, writeOut=False)
            assert abs(ev[0] - g["ev"][0]) <= 1e-10 * abs(g["ev"][0])        # north-star tolerance
            assert st["cumIter"] == int(g["cumIter"]) and st["isConverged"]
            assert isinstance(ev, np.ndarray) and isinstance(Y, list) and isinstance(Y[0], hip.HipVector)
            ov = abs(np.dot(Y[0].array, g["vec0"]))
            assert abs(ov - 1) < 1e-8
            S = hip.HipVector.overlapMatrix(Y)
            np.testing.assert_allclose(S, np.eye(len(Y)), atol=1e-7)
            res = hip.true_residual_norms(H, ev, Y, 1)
            assert res[0] < 1e-6                 # ||H y - theta y||, eigenvalue converged to 1e-13


def test_lanczos_checkpoint_resume_and_thick_restart_on_device(hip, gapped4000, tmp_path):
    """Resuming from a mid-run checkpoint continues exactly where the run stood: the kernels used at
    this size have a fixed summation order, so the resumed run is bit-identical."""
    Hh, guess = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    run = lambda **kw: hip.inexactLanczosDiagonalization(H, hip.HipVector(guess.copy(), _opts()), 0.02, 5, 6, 1e-12,
                                                         writeOut=False, **kw)
    d = str(tmp_path / "ck")
    ev, Y, st = run(checkpointDir=d, checkpointKeep=0)
    assert st["isConverged"] and st["cumIter"] > 5
    # not only self-consistent: the CPU restatement of the reference loop at the same parameters (L = 5, no golden
    # file covers them) gives the same Ritz value, iteration count and exit
    evo, Yo, sto = lanczos_ref.inexact_lanczos(Hh, RefVector(guess.copy(), _opts()), 0.02, 5, 6, 1e-12)
    assert abs(ev[0] - evo[0]) <= 1e-10 * abs(evo[0]) and st["cumIter"] == sto["cumIter"] and sto["isConverged"]
    for it in (2, 4, 5):
        ev2, Y2, st2 = run(resumeFrom=f"{d}/krylov_{it:06d}.npz")
        np.testing.assert_array_equal(ev2, ev)
        assert st2["cumIter"] == st["cumIter"] and st2["residual"] == st["residual"]
        assert isinstance(Y2[0], hip.HipVector)
        np.testing.assert_array_equal(Y2[0].array, Y[0].array)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev_t, Y_t, st_t = run(thickRestart=2)
    assert st_t["isConverged"] and abs(ev_t[0] - ev[0]) <= 1e-10 * abs(ev[0])


def test_lanczos_dense_reference_test_case(hip):
    """unittests/test_lanczos.py restated for HipVector with MINRES as the inner solver (the dense matrix stored
    as full CSR); the like-for-like gcrotmk run is test_reference_unit_tests_with_gcrotmk_on_device."""
    g = load_golden("lanczos_n100_seed1212.npz")
    A, exact = dense_test_matrix(100, 1212)
    H = hip.HipCsrOperator.from_dense(A)
    ev, Y, st = hip.inexactLanczosDiagonalization(H, hip.HipVector(g["guess"].copy(), _opts(1000, 1e-4)),
                                                  30, 6, 4, 1e-6, writeOut=False)
    evo, Yo, sto = lanczos_ref.inexact_lanczos(A, RefVector(g["guess"].copy(), _opts(1000, 1e-4)), 30, 6, 4, 1e-6)
    assert st["cumIter"] == sto["cumIter"] and st["isConverged"] == sto["isConverged"]
    np.testing.assert_allclose(ev[0], evo[0], rtol=1e-5)       # inexact solves (tol 1e-4), eConv 1e-6
    assert abs(hip.find_nearest(ev, 30)[1] - hip.find_nearest(exact, 30)[1]) <= 1e-4
    S = hip.HipVector.overlapMatrix(Y)
    np.testing.assert_allclose(S, np.eye(len(Y)), atol=1e-5)
    S1 = hip.HipVector.overlapMatrix(Y[:-1])
    np.testing.assert_allclose(hip.HipVector.extendOverlapMatrix(Y, S1), S, atol=1e-9)
    w, V = np.linalg.eigh(A)
    vec = Y[hip.find_nearest(ev, 30)[0]].array
    ov = np.vdot(V[:, hip.find_nearest(w, 30)[0]], vec)
    np.testing.assert_allclose(abs(ov), 1, rtol=1e-5)


def test_lanczos_block_and_lindep_exit_match_reference(hip, gapped4000):
    Hh, _ = gapped4000
    H = hip.HipCsrOperator.from_scipy(Hh)
    for nb, L, maxit, tol, econv, tag in ((3, 3, 12, 1e-8, 1e-7, "block3"), (4, 3, 12, 1e-10, 1e-7, "block4_lindep")):
        g = load_golden(f"gapped_csr_n4000_{tag}.npz")
        Q = la.qr(np.random.default_rng(5).standard_normal((4000, nb)), mode="economic")[0]
        v0 = [hip.HipVector(Q[:, i].copy(), _opts(2000, tol, orthogonalization="mgs")) for i in range(nb)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev, Y, st = hip.inexactLanczosDiagonalization(H, v0, 0.02, L, maxit, econv, writeOut=False)
        if tag == "block3":
            np.testing.assert_allclose(np.sort(ev[:nb]), np.sort(g["ev"][:nb]), rtol=1e-6)   # eConv = 1e-7 run
            assert st["isConverged"] and st["cumIter"] == int(g["cumIter"])
        else:
            assert np.all(np.isnan(ev)) and not st["isConverged"]
            assert st["cumIter"] == int(g["cumIter"]) and len(Y) == int(g["nvec"])
    bad = [hip.HipVector(np.ones(4000), _opts()), hip.HipVector(np.ones(4000), _opts())]
    with pytest.raises(RuntimeError):
        hip.inexactLanczosDiagonalization(H, bad, 0.02, 3, 1, 1e-6, writeOut=False)


# ---------------------------------------------------------------- full-size properties
def test_full_size_operator_properties(hip):
    """N = 1e6, nnz/row = 32 (BASELINE config #2 shape): size-independent properties -
    symmetry <x,Hy> = <Hx,y>, linearity, agreement of the two kernel variants, the shifted
    form, and a slab generated separately equals the same rows of the full operator."""
    N = 1_000_000
    H = hip.HipCsrOperator.generate(N, 32, seed=7)
    assert abs(H.nnz / N - 33) < 0.1
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(N), rng.standard_normal(N)
    X, Y = hip.HipVector(x), hip.HipVector(y)
    HX, HY = X.applyOp(H), Y.applyOp(H)
    a, b = X.vdot(HY), HX.vdot(Y)
    assert abs(a - b) <= 1e-12 * (abs(a) + np.sqrt(N))
    Z = hip.HipVector.linearCombination([X, Y], [2.0, -3.0])
    lhs = Z.applyOp(H)
    rhs = hip.HipVector.linearCombination([HX, HY], [2.0, -3.0])
    d = hip.HipVector.linearCombination([lhs, rhs], [1.0, -1.0])
    assert d.norm() <= 1e-14 * rhs.norm() * 10
    for variant in (1, 2, 3, 4):                     # CSR-vector, CSR-stream, column-window blocked
        H.set_variant(variant)
        hv = X.applyOp(H)
        d = hip.HipVector.linearCombination([hv, HX], [1.0, -1.0])
        assert d.norm() <= 1e-14 * HX.norm()
    H.set_variant(0)
    # rows 400000..400500 against the host generator's slab
    slab = gapped_csr_host(N, 32, seed=7, row_begin=400000, row_end=400500)
    _within(HX.array[400000:400500], slab @ x, 1e-13 * (np.abs(slab) @ np.abs(x)))
    # reproducibility: variants 1-3 fix the summation order (bitwise equal runs); variant 4 adds a row
    # from several waves, so two runs may differ in the last bits only
    for variant, bitwise in ((2, True), (3, True), (4, False)):
        H.set_variant(variant)
        r1, r2 = X.applyOp(H).array, X.applyOp(H).array
        if bitwise:
            assert np.array_equal(r1, r2)
        else:
            assert np.max(np.abs(r1 - r2)) <= 1e-15 * np.max(np.abs(r1)) * 8
    assert X.vdot(Y) == X.vdot(Y) and X.norm() == X.norm()          # reductions: fixed tree
    # two operators with different LDS footprints used alternately (kernel attributes are global)
    small_h = gapped_csr_host(4000, 32, seed=7)
    small = hip.HipCsrOperator.from_scipy(small_h)
    xs = rng.standard_normal(4000)
    for variant in (4, 3):
        small.set_variant(variant); H.set_variant(variant)
        for _ in range(2):
            ys = hip.HipVector(xs).applyOp(small).array
            _within(ys, small_h @ xs, 1e-13 * (np.abs(small_h) @ np.abs(xs)))
            d = hip.HipVector.linearCombination([X.applyOp(H), HX], [1.0, -1.0])
            assert d.norm() <= 1e-14 * HX.norm()
    H.set_variant(0)
    buf = hip.HipContext.default().alloc(N)
    H.apply_shifted(0.02, X._buf, buf)
    s = hip.HipVector(buf)
    t = hip.HipVector.linearCombination([X, HX], [0.02, -1.0])
    d = hip.HipVector.linearCombination([s, t], [1.0, -1.0])
    assert d.norm() <= 1e-15 * t.norm() * 10


def test_state_following_by_max_overlap_on_device(hip):
    """unittests/test_stateFollowingHO.py with HipVector (MINRES as inner solver)."""
    from eigensolvers_amd.generators import sinc_dvr_harmonic
    Hd, _ = sinc_dvr_harmonic(45, (-10, 10))
    w, V = la.eigh(Hd)
    sigma = 13.1
    idx = hip.find_nearest(w, sigma)[0]
    opts = _opts(30000, 1e-4)
    ref = hip.HipVector(V[:, idx + 1].copy(), opts)
    np.random.seed(13)
    y0 = hip.HipVector(np.random.random(45), opts)
    ev, Y, st = hip.inexactLanczosDiagonalization(hip.HipCsrOperator.from_dense(Hd), y0, sigma, 16, 200, 1e-10,
                                                  pick=hip.get_pick_function_maxOvlp(ref), writeOut=False)
    assert st["isConverged"]
    assert abs(ev[0] - w[idx + 1]) / w[idx + 1] <= 1e-4
    np.testing.assert_allclose(abs(np.vdot(ref.array, Y[0].array)), 1, rtol=1e-2)


# ---------------------------------------------------------------- GCROT(m,k) on the device
@pytest.mark.parametrize("cols", [1, 4])
@pytest.mark.parametrize("n,m", [(1000, 0), (1000, 1), (100003, 5), (100002, 0), (100002, 4), (100002, 9), (1 << 20, 40),
                                 (3_000_001, 7)])
def test_arnoldi_step_against_numpy(hip, n, m, cols):
    """hipeig_arnoldi_step / hipeig_pair_arnoldi_step (norm, sequential MGS, norm, scaling in one call)
    against NumPy's evaluation of the same sequence, real and complex-as-pairs.  cols = 4: the blocked form
    (hipeig_arnoldi_step_p: four columns per pass, sequential coefficients recovered through the block's Gram matrix) -
    the same algebra, so the same expectations; the columns here are NOT orthonormal, the Gram terms matter."""
    from eigensolvers_amd.gcrotmk import _Ops, _PairOps
    rng = np.random.default_rng(n + m)
    Vh = rng.standard_normal((m, n)) / np.sqrt(n)
    Vih = rng.standard_normal((m, n)) / np.sqrt(n)
    if m >= 2:
        Vh[1] += 0.3 * Vh[0]                               # a pair of columns far from orthogonal
    wh, wih = rng.standard_normal(n), rng.standard_normal(n)
    ctx = hip.HipContext.default()
    ops, pops = _Ops(ctx, n, cols), _PairOps(ctx, n, cols)
    V = [hip.HipVector(Vh[j].copy())._buf for j in range(m)]
    w = hip.HipVector(wh.copy())
    nb_d, h_d, na_d = ops.arnoldi_step(V, w._buf)
    wv = wh.copy(); h = []
    for j in range(m):
        c = Vh[j] @ wv; h.append(c); wv -= c * Vh[j]
    nb, na = np.linalg.norm(wh), np.linalg.norm(wv)
    assert abs(nb_d - nb) <= 1e-13 * nb and abs(na_d - na) <= 1e-12 * nb
    np.testing.assert_allclose(h_d, np.array(h), rtol=0, atol=1e-12 * nb)
    np.testing.assert_allclose(w.array, wv / na, rtol=0, atol=1e-12)
    Vp = [(hip.HipVector(Vh[j].copy())._buf, hip.HipVector(Vih[j].copy())._buf) for j in range(m)]
    wr, wi = hip.HipVector(wh.copy()), hip.HipVector(wih.copy())
    nb_d, h_d, na_d = pops.arnoldi_step(Vp, (wr._buf, wi._buf))
    z = wh + 1j * wih; hz = []
    for j in range(m):
        v = Vh[j] + 1j * Vih[j]
        c = np.vdot(v, z); hz.append(c); z = z - c * v
    nz = np.linalg.norm(wh + 1j * wih)
    assert abs(nb_d - nz) <= 1e-13 * nz and abs(na_d - np.linalg.norm(z)) <= 1e-12 * nz
    np.testing.assert_allclose(h_d, np.array(hz, dtype=complex), rtol=0, atol=1e-12 * nz)
    np.testing.assert_allclose(wr.array + 1j * wi.array, z / np.linalg.norm(z), rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [37, 4000, 8192])
def test_batched_arnoldi_steps_equal_the_single_steps(hip, n):
    """hipeig_pair_arnoldi_step_batch_begin: the orthogonalisation steps of several right-hand sides (each against its OWN
    columns, different counts, one with none) in one launch, a workgroup per step - the same device code as the single
    step at these lengths, so coefficients, norms and the normalised vectors are bit-identical; a longer vector is
    declined (the caller then enqueues the steps one by one)."""
    from eigensolvers_amd.gcrotmk import _PairOps
    import os
    if os.environ.get("HIPEIG_MAPPED_SCALARS") == "0":
        pytest.skip("the batched form needs the mapped scalar area (it is declined without it: status 5)")
    ctx = hip.HipContext.default()
    rng = np.random.default_rng(n)
    counts = [0, 1, 7, 40, 62, 3]

    def make():
        r = np.random.default_rng(n + 1)
        reqs = []
        for m in counts:
            vs = [(hip.HipVector(r.standard_normal(n) / np.sqrt(n))._buf, hip.HipVector(r.standard_normal(n) / np.sqrt(n))._buf)
                  for _ in range(m)]
            reqs.append((vs, (hip.HipVector(r.standard_normal(n))._buf, hip.HipVector(r.standard_normal(n))._buf)))
        return reqs

    opss = [_PairOps(ctx, n) for _ in counts]
    single = make()
    ref = [opss[i].arnoldi_step(vs, w) for i, (vs, w) in enumerate(single)]
    batch = make()
    assert _PairOps.arnoldi_begin_batch(opss, batch)
    for slot, (vs, w) in enumerate(batch):
        nb, h, na = opss[slot].arnoldi_end(len(vs), slot)
        assert nb == ref[slot][0] and na == ref[slot][2]
        np.testing.assert_array_equal(h, ref[slot][1])
        for part, part_ref in zip(w, single[slot][1]):
            np.testing.assert_array_equal(hip.HipVector(part).array, hip.HipVector(part_ref).array)
    big = _PairOps(ctx, 8193)
    wbig = (hip.HipVector(rng.standard_normal(8193))._buf, hip.HipVector(rng.standard_normal(8193))._buf)
    assert _PairOps.arnoldi_begin_batch([big, big], [([], wbig), ([], wbig)]) is False


def test_gcrotmk_tracks_scipy(hip, gapped4000):
    """linearSolver="gcrotmk" (numpyVector.py:161): device GCROT(20,20) against the SciPy routine
    the reference calls, on the shifted gapped operator and on the reference's dense test matrix."""
    import scipy.sparse.linalg as spla
    Hh, guess = gapped4000
    b = guess / np.linalg.norm(guess)
    H = hip.HipCsrOperator.from_scipy(Hh)
    for rtol in (1e-4, 1e-10):
        op = spla.LinearOperator(Hh.shape, matvec=lambda x: 0.02 * x - Hh @ x, dtype=np.float64)
        xs, info = spla.gcrotmk(op, b, rtol=rtol, atol=1e-14, maxiter=2000)
        opts = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 2000, "linear_tol": rtol,
                                     "linear_atol": 1e-14}}
        W = hip.HipVector.solve(H, hip.HipVector(b.copy(), opts), 0.02)
        assert info == 0
        # both satisfy ||b - A x|| <= rtol ||b||, so they agree to cond(A) * rtol (|sigma - lambda|_min ~ 7e-3)
        _within(W.array, xs, max(1e-9, 300 * rtol) * np.linalg.norm(xs))
        r = b - (0.02 * W.array - Hh @ W.array)
        assert np.linalg.norm(r) <= rtol * np.linalg.norm(b) * 1.0001
    with pytest.raises(UserWarning):                              # not converged -> raises like the reference
        hip.HipVector.solve(H, hip.HipVector(b.copy(), {"linearSystemArgs": {
            "linearSolver": "gcrotmk", "linearIter": 1, "linear_tol": 1e-12, "linear_atol": 0.0}}), 0.02)


def test_reference_unit_tests_with_gcrotmk_on_device(hip):
    """unittests/test_lanczos.py and test_lanczosBlock.py exactly as the reference runs them
    (gcrotmk, tol 1e-4), with HipVector: same iteration counts, eigenvalues to the tests' bounds."""
    opt = lambda: {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-4}}
    g = load_golden("lanczos_n100_seed1212.npz")
    A, exact = dense_test_matrix(100, 1212)
    ev, Y, st = hip.inexactLanczosDiagonalization(hip.HipCsrOperator.from_dense(A), hip.HipVector(g["guess"].copy(), opt()),
                                                  30, 6, 4, 1e-6, writeOut=False)
    assert st["cumIter"] == int(g["cumIter"]) and st["isConverged"]
    np.testing.assert_allclose(ev[:2], g["ev"], rtol=1e-6)
    assert abs(hip.find_nearest(ev, 30)[1] - hip.find_nearest(exact, 30)[1]) <= 1e-4
    # the remaining sub-tests of test_lanczos.py:44-91 on the returned device vectors
    from eigensolvers_amd.subspace import loewdin_transform, ritz_pairs
    Hd = hip.HipCsrOperator.from_dense(A)
    assert isinstance(ev, np.ndarray) and isinstance(Y, list) and isinstance(Y[0], hip.HipVector)          # "returnType"
    S = hip.HipVector.overlapMatrix(Y)
    np.testing.assert_allclose(S, np.eye(len(Y)), atol=1e-5)                                               # "orthogonal"
    Hm = hip.HipVector.matrixRepresentation(Hd, Y)
    uS = loewdin_transform(S)[1]
    uSH = uS @ ritz_pairs(uS, Hm)[1]
    np.testing.assert_allclose(uSH.T.conj() @ S @ uSH, np.eye(uSH.shape[1]), atol=1e-5)                    # "transformationMatrix"
    np.testing.assert_allclose(hip.HipVector.extendOverlapMatrix(Y, hip.HipVector.overlapMatrix(Y[:-1])), S, atol=1e-9)   # "extension"
    np.testing.assert_allclose(hip.HipVector.extendMatrixRepresentation(Hd, Y, hip.HipVector.matrixRepresentation(Hd, Y[:-1])),
                               Hm, atol=1e-9)
    w1, V1 = np.linalg.eigh(A)
    exact_vec = V1[:, hip.find_nearest(w1, 30)[0]]                                                         # "eigenvector"
    lan_vec = Y[hip.find_nearest(ev, 30)[0]].array
    ov = np.vdot(exact_vec, lan_vec)
    np.testing.assert_allclose(abs(ov), 1, rtol=1e-5)
    np.testing.assert_allclose(exact_vec, lan_vec * ov, rtol=1e-5, atol=1e-4)
    gb = load_golden("block3_degenerate.npz")
    Ab, _ = dense_test_matrix(100, 1212, gb["exact"])
    v0 = [hip.HipVector(gb["guess"][:, i].copy(), opt()) for i in range(3)]
    ev, Y, st = hip.inexactLanczosDiagonalization(hip.HipCsrOperator.from_dense(Ab), v0, gb["exact"][5] + 1.5, 6, 4, 1e-6,
                                                  writeOut=False)
    assert st["cumIter"] == int(gb["cumIter"])
    np.testing.assert_allclose(ev[:3], gb["exact"][5:8], rtol=1e-6)          # test_lanczosBlock.py:54
    w, V = np.linalg.eigh(Ab)
    lan = np.vstack([Y[i].array for i in range(3)]).T
    assert abs(np.abs(la.eigvals(lan.T @ V[:, 5:8])).sum() - 3) < 1e-6       # projector trace, :56-62


def test_reference_lindep_test_case_on_device(hip):
    """unittests/test_lanczosLINDEP.py with HipVector (dense n = 1200 operator, gcrotmk rtol 1e-1, sigma = 390, L = 100,
    maxit = 1000), against the reference's own two runs (tests/golden/lindep_dense_n1200.npz).  The solves are
    deliberately sloppy, so the comparison is on what the reference's test reads: the status fields and the count of
    returned vectors; the converged Ritz values agree to the accuracy the run itself reached."""
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_golden_r2 import lindep_case, LINDEP
    g = load_golden("lindep_dense_n1200.npz")
    A, y0 = lindep_case()
    H = hip.HipCsrOperator.from_dense(A)
    opt = lambda: {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": LINDEP["linearIter"], "linear_tol": LINDEP["linear_tol"]}}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = hip.inexactLanczosDiagonalization(H, hip.HipVector(y0.copy(), opt()), LINDEP["sigma"], LINDEP["L"], LINDEP["maxit"],
                                                      1e-12, writeOut=False)
    assert st["isConverged"] and not st["lindep"] and st["outerIter"] == 0 and st["futileRestarts"] == 0
    assert abs(st["cumIter"] - int(g["cumIter_a"])) <= 2 and len(Y) == st["cumIter"] + 1
    np.testing.assert_allclose(ev[:3], g["ev_a"][:3], rtol=1e-9)
    exact = np.linspace(1, 400, 1200)
    assert abs(ev[0] - exact[np.argmin(np.abs(exact - 390))]) < 1e-6
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = hip.inexactLanczosDiagonalization(H, hip.HipVector(y0.copy(), opt()), LINDEP["sigma"], LINDEP["L"], LINDEP["maxit"],
                                                      1e-18, writeOut=False)
    # the reference: cycle exhausted (cumIter = L), one restart, NaN Ritz values, one vector back
    assert (st["outerIter"], st["cumIter"], st["isConverged"]) == (int(g["outerIter_b"]), int(g["cumIter_b"]), False)
    assert np.all(np.isnan(ev)) and len(Y) == int(g["nvec_b"]) == st["innerIter"]


def test_solve_with_an_initial_guess_tracks_scipy(hip, gapped4000):
    """``solve(H, b, sigma, x0)``: NumpyVector hands x0 on to SciPy (numpyVector.py:161,163).  The reference's solvers
    never pass one, so there is no golden run; the check is against scipy.sparse.linalg itself on the same operator
    (the third-party routine the reference calls): MINRES iteration count equal, iterates to the solve tolerance;
    an exact x0 is returned at once; GCROT converges from x0 to the same solution."""
    import scipy.sparse.linalg as spla
    Hh, guess = gapped4000
    n = Hh.shape[0]
    H = hip.HipCsrOperator.from_scipy(Hh)
    rng = np.random.default_rng(21)
    b = guess / np.linalg.norm(guess)
    x0 = 0.3 * rng.standard_normal(n)
    lin = spla.LinearOperator((n, n), matvec=lambda v: 0.02 * v - Hh @ v, dtype=np.float64)
    for rtol in (1e-6, 1e-10):
        its = []
        ref, info = spla.minres(lin, b, x0, rtol=rtol, maxiter=2000, callback=lambda xk: its.append(1))
        assert info == 0
        W = hip.HipVector.solve(H, hip.HipVector(b.copy(), _opts(2000, rtol)), 0.02, hip.HipVector(x0.copy()))
        assert W.last_solve_stats["iterations"] == len(its)
        assert np.linalg.norm(W.array - ref) <= max(1e-9, 10 * rtol) * np.linalg.norm(ref)
        W2 = hip.HipVector.solve(H, hip.HipVector(b.copy(), _opts(2000, rtol)), 0.02, x0.copy())       # ndarray guess, as NumpyVector takes it
        np.testing.assert_array_equal(W2.array, W.array)
    # reverseGF flips the operator's sign (numpyVector.py:153-154)
    linr = spla.LinearOperator((n, n), matvec=lambda v: Hh @ v - 0.02 * v, dtype=np.float64)
    ref, info = spla.minres(linr, b, x0, rtol=1e-8, maxiter=2000)
    W = hip.HipVector.solve(H, hip.HipVector(b.copy(), _opts(2000, 1e-8)), 0.02, hip.HipVector(x0.copy()), reverseGF=True)
    assert np.linalg.norm(W.array - ref) <= 1e-7 * np.linalg.norm(ref)
    # an exact initial guess: beta1 = 0, x0 comes back untouched, no iteration
    xe = hip.HipVector(rng.standard_normal(n))
    buf = hip.HipContext.default().alloc(n)
    H.apply_shifted(0.02, xe._buf, buf)                         # b = A xe as the device evaluates it: r1 = b - A xe = 0 exactly
    We = hip.HipVector.solve(H, hip.HipVector(buf, _opts(2000, 1e-10)), 0.02, xe)
    assert We.last_solve_stats["iterations"] == 0
    np.testing.assert_array_equal(We.array, xe.array)
    # GCROT from x0
    og = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-9, "linear_atol": 1e-12}}
    refg, infog = spla.gcrotmk(lin, b, x0, rtol=1e-9, atol=1e-12, maxiter=1000)
    Wg = hip.HipVector.solve(H, hip.HipVector(b.copy(), og), 0.02, hip.HipVector(x0.copy()))
    assert infog == 0 and np.linalg.norm(Wg.array - refg) <= 1e-7 * np.linalg.norm(refg)
    assert np.linalg.norm(0.02 * Wg.array - Hh @ Wg.array - b) <= 2e-9

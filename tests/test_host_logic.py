"""Host-side logic of the product on CPU: the backend-agnostic Lanczos driver (run here on
the oracle's ndarray vector, the device backend being unavailable without a GPU), the
subspace helpers, the synthetic generator and the partitioning helpers."""
import os
import warnings

import numpy as np
import pytest
import scipy.linalg as la

import eigensolvers_amd as ea
from conftest import GOLDEN, load_golden
from eigensolvers_amd import subspace
from eigensolvers_amd.distributed import all_row_ranges, row_range
from eigensolvers_amd.generators import dense_test_matrix, gapped_csr_host, gapped_params
from oracle.numpy_vector import RefVector

ea.AbstractVector.register(RefVector)


def _opts(solver, it, tol):
    return {"linearSystemArgs": {"linearSolver": solver, "linearIter": it, "linear_tol": tol}}


def test_driver_reproduces_reference_dense_case():
    g = load_golden("lanczos_n100_seed1212.npz")
    A, exact = dense_test_matrix(100, 1212)
    ev, Y, st = ea.inexactLanczosDiagonalization(A, RefVector(g["guess"].copy(), _opts("gcrotmk", 1000, 1e-4)),
                                                 30, 6, 4, 1e-6, writeOut=False)
    np.testing.assert_allclose(ev, g["ev"], rtol=1e-6)
    assert st["cumIter"] == int(g["cumIter"]) and st["isConverged"]
    # the reference test's own checks (unittests/test_lanczos.py:49-93)
    assert isinstance(ev, np.ndarray) and isinstance(Y, list) and isinstance(Y[0], RefVector)
    S = RefVector.overlapMatrix(Y)
    np.testing.assert_allclose(S, np.eye(len(Y)), atol=1e-5)
    S1 = RefVector.overlapMatrix(Y[:-1])
    np.testing.assert_allclose(RefVector.extendOverlapMatrix(Y, S1), S, atol=1e-9)
    H1 = RefVector.matrixRepresentation(A, Y[:-1])
    np.testing.assert_allclose(RefVector.extendMatrixRepresentation(A, Y, H1),
                               RefVector.matrixRepresentation(A, Y), atol=1e-9)
    assert abs(ea.find_nearest(ev, 30)[1] - ea.find_nearest(exact, 30)[1]) <= 1e-4
    w, V = np.linalg.eigh(A)
    vec = Y[ea.find_nearest(ev, 30)[0]].array
    ov = np.vdot(V[:, ea.find_nearest(w, 30)[0]], vec)
    np.testing.assert_allclose(abs(ov), 1, rtol=1e-5)
    np.testing.assert_allclose(V[:, ea.find_nearest(w, 30)[0]], vec * ov, rtol=1e-5, atol=1e-4)


def test_driver_equals_oracle_loop_on_sparse_case(gapped4000):
    H, guess = gapped4000
    g = load_golden("gapped_csr_n4000_minres.npz")
    ev, Y, st = ea.inexactLanczosDiagonalization(H, RefVector(guess.copy(), _opts("minres", 2000, 1e-10)),
                                                 0.02, 8, 10, 1e-13, writeOut=False)
    assert abs(ev[0] - g["ev"][0]) <= 1e-11 * abs(g["ev"][0])
    assert st["cumIter"] == int(g["cumIter"]) and st["isConverged"]
    for key in ("ref", "residual", "nBlock", "flagAddition", "outerIter", "innerIter", "cumIter", "iBlock",
                "zeroVector", "isConverged", "lindep", "futileRestarts", "startTime", "runTime", "phase"):
        assert key in st                                   # inexact_Lanczos.py:65-73


def test_driver_block_and_lindep_exit(gapped4000):
    H, _ = gapped4000
    for nb, L, maxit, tol, econv, tag in ((3, 3, 12, 1e-8, 1e-7, "block3"), (4, 3, 12, 1e-10, 1e-7, "block4_lindep")):
        g = load_golden(f"gapped_csr_n4000_{tag}.npz")
        Q = la.qr(np.random.default_rng(5).standard_normal((4000, nb)), mode="economic")[0]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev, Y, st = ea.inexactLanczosDiagonalization(
                H, [RefVector(Q[:, i].copy(), _opts("minres", 2000, tol)) for i in range(nb)],
                0.02, L, maxit, econv, writeOut=False)
        np.testing.assert_allclose(ev, g["ev"], rtol=1e-7, equal_nan=True)
        assert st["cumIter"] == int(g["cumIter"]) and st["isConverged"] == bool(g["isConverged"])
        assert len(Y) == int(g["nvec"])


def test_driver_argument_errors_and_guess_normalisation():
    A, _ = dense_test_matrix(20, 3)
    bad = [RefVector(np.ones(20), {}), RefVector(np.ones(20), {})]
    with pytest.raises(RuntimeError):                      # inexact_Lanczos.py:289-291
        ea.inexactLanczosDiagonalization(A, bad, 30, 3, 1, 1e-6, writeOut=False)
    v = RefVector(np.full(20, 2.0), _opts("gcrotmk", 100, 1e-6))
    ea.inexactLanczosDiagonalization(A, v, 30, 3, 1, 1e-6, writeOut=False)
    assert abs(np.linalg.norm(v.array) - 1) < 1e-12        # caller's guess normalised in place (:294)
    with pytest.raises(UserWarning):                       # non-converged inner solve raises
        ea.inexactLanczosDiagonalization(A, RefVector(np.ones(20), _opts("minres", 1, 1e-14)), 30, 3, 1, 1e-6,
                                         writeOut=False)


def _summary_rows(text):
    """(frame lines, table rows) of a summary file: dates and the wall-clock column masked."""
    frame, rows = [], []
    for line in text.splitlines():
        cells = line.split()
        if cells and cells[0].isdigit() and len(cells) >= 7:
            rows.append(cells[:-1])                         # drop time(seconds)
        elif "/" in line and ":" in line and line.startswith("\t\t"):
            frame.append("<date>")
        else:
            frame.append(line)
    return frame, rows


@pytest.mark.parametrize("case", ["single", "block8"])
def test_summary_file_equals_the_references(tmp_path, gapped4000, case):
    """f3: the summary file against the one the reference's LanczosPrintUtils wrote for the same run
    (printUtils.py:59-187, 249-270; tests/golden/summary_lanczos_*.out from make_golden_r2.py): same
    frame, same column layout, same iteration counters, same printed eigenvalues and residuals."""
    H, guess = gapped4000
    out, summ = tmp_path / "it.out", tmp_path / "sum.out"
    if case == "single":
        v0 = RefVector(guess.copy(), {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000,
                                                           "linear_tol": 1e-10, "linear_atol": 1e-12}})
        args, kw = (0.02, 8, 10, 1e-13), dict(eShift=0.005)
    else:
        g = load_golden("gapped_csr_n4000_block8.npz")
        Q = la.qr(np.random.default_rng(5).standard_normal((4000, 8)), mode="economic")[0]
        v0 = [RefVector(Q[:, i].copy(), _opts("minres", 2000, float(g["linear_tol"]))) for i in range(8)]
        args, kw = (0.02, int(g["L"]), int(g["maxit"]), float(g["eConv"])), {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = ea.inexactLanczosDiagonalization(H, v0, *args, writeOut=True, outFileName=str(out),
                                                     summaryFileName=str(summ), **kw)
    ref_text = open(os.path.join(GOLDEN, f"summary_lanczos_{case}.out")).read()
    frame, rows = _summary_rows(summ.read_text())
    rframe, rrows = _summary_rows(ref_text)
    assert frame == rframe                                   # banner, parameter block, column header, markers
    assert frame[0] == "startingPoint" and "endingPoint" in frame
    assert len(rows) == len(rrows) == st["cumIter"]
    for r, rr in zip(rows, rrows):
        assert r[:4] == rr[:4]                               # it, i, nCum, target
        if r != rr:                                          # another BLAS: the printed digits may move
            np.testing.assert_allclose([float(c) for c in r[4:-1]], [float(c) for c in rr[4:-1]], rtol=0, atol=5e-6)
    if case == "single":
        tail = out.read_text().split("FINAL RESULTS")[-1].split("*" * 70)[0]
        rtail = open(os.path.join(GOLDEN, "iterations_lanczos_single_tail.out")).read().split("*" * 70)[0]
        assert tail.split() == rtail.split()                 # eigenvalue block and the "Target, Lanczos (nearest)" line


def test_subspace_helpers_against_reference_outputs(gapped4000):
    g = load_golden("subspace_helpers.npz")
    gm = load_golden("gram_n4000.npz")
    ok, X = subspace.loewdin_transform(gm["S"])
    assert ok
    np.testing.assert_allclose(X, g["uS"], rtol=1e-10, atol=1e-12)
    theta, _ = subspace.ritz_pairs(X, gm["Hm"])
    np.testing.assert_allclose(theta, g["evs"], rtol=1e-11)
    assert subspace.eigenvalue_change(np.array([1.0, 2.0, 3.5]), np.array([1.1, 1.9, 3.0])) == float(g["resid"])
    ok, X = subspace.loewdin_transform(np.array([[1.0, 1.0], [1.0, 1.0]]))
    assert not ok and X.shape == (2, 1)                    # dependent direction dropped (util_funcs.py:241-243)
    np.testing.assert_array_equal(subspace.get_pick_function_close_to_sigma(2.0)(None, None, np.array([5.0, 1.9, 2.5])),
                                  [1, 2, 0])
    vs = [RefVector(np.eye(3)[i], {}) for i in range(3)]
    pick = subspace.get_pick_function_maxOvlp(RefVector(np.array([0.1, 0.9, 0.3]), {}))
    np.testing.assert_array_equal(pick(np.eye(3), vs, None), [1, 2, 0])
    # basisTransformation incl. the [1.0] quirk (util_funcs.py:224-225)
    assert subspace.basisTransformation(vs[:1], np.array([1.0]))[0] is not None
    assert isinstance(subspace.basisTransformation(vs[:1], np.array([1.0]))[0], list)
    out = subspace.basisTransformation(vs, np.array([[1.0, 0.0], [0.0, 2.0], [3.0, 0.0]]))
    np.testing.assert_array_equal(out[0].array, [1, 0, 3])
    np.testing.assert_array_equal(out[1].array, [0, 2, 0])


def test_generator_properties():
    H = gapped_csr_host(3001, 32, seed=11)                  # odd N exercises cycle walking
    assert abs(H - H.T).max() == 0.0
    lens = np.diff(H.indptr)
    assert 24 <= lens.min() and lens.max() <= 42 and abs(lens.mean() - 33) < 0.5
    assert np.all(np.diff(H.indices[H.indptr[5]:H.indptr[6]]) >= 0)       # rows sorted by column
    ev = np.linalg.eigvalsh(H.toarray())
    inside = ev[np.abs(ev) < 0.5]
    assert len(inside) == 16                                # the mid-spectrum cluster, gap to +-0.9
    np.testing.assert_allclose(np.sort(inside), np.linspace(-0.2, 0.2, 16), atol=2e-3)
    slab = gapped_csr_host(3001, 32, seed=11, row_begin=700, row_end=1900)
    assert abs(slab - H[700:1900]).max() == 0.0 and np.array_equal(slab.indices, H[700:1900].indices)
    assert abs(gapped_csr_host(3001, 32, seed=12) - H).max() > 0
    p = gapped_params(10**7, 64)
    assert p["K"] == 36 and abs(2 * p["K"] * p["thresh24"] / 2**24 - 64) < 1e-4


def test_row_ranges_tile_the_rows():
    for N, P in ((10, 3), (10**7, 8), (7, 8), (4000, 2)):
        rr = all_row_ranges(N, P)
        assert rr[0][0] == 0 and rr[-1][1] == N
        assert all(rr[i][1] == rr[i + 1][0] for i in range(P - 1))
        assert max(e - b for b, e in rr) - min(e - b for b, e in rr) <= 1
        assert row_range(N, P, P - 1) == rr[-1]


def test_product_never_imports_the_oracle():
    import eigensolvers_amd
    root = os.path.dirname(eigensolvers_amd.__file__)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_state_following_by_max_overlap():
    """unittests/test_stateFollowingHO.py restated (the in-house `basis.SincInfInf` is replaced by
    the Colbert-Miller sinc-DVR of generators.sinc_dvr_harmonic): follow the state above the one
    nearest to sigma by maximum overlap."""
    from eigensolvers_amd.generators import sinc_dvr_harmonic
    H, _ = sinc_dvr_harmonic(45, (-10, 10))
    w, V = la.eigh(H)
    np.testing.assert_allclose(w[:8], 2 * np.arange(8) + 1, rtol=1e-8)         # harmonic ladder
    sigma = 13.1
    idx = ea.find_nearest(w, sigma)[0]
    opts = _opts("gcrotmk", 30000, 1e-4)
    ref = RefVector(V[:, idx + 1].copy(), opts)
    np.random.seed(13)
    y0 = RefVector(np.random.random(45), opts)
    ev, Y, st = ea.inexactLanczosDiagonalization(H, y0, sigma, 16, 200, 1e-10,
                                                 pick=ea.get_pick_function_maxOvlp(ref), writeOut=False)
    assert st["isConverged"]
    assert abs(ev[0] - w[idx + 1]) / w[idx + 1] <= 1e-4
    np.testing.assert_allclose(abs(np.vdot(ref.array, Y[0].array)), 1, rtol=1e-2)

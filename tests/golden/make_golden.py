"""Generate tests/golden/*.npz by running the REAL reference implementation.

Runs only in the build container (needs /root/reference); its outputs are committed so
that nothing at test / GPU time ever reads the reference tree.  It imports the upstream
modules unchanged through the harness of SURVEY.md section 8c:

* stub modules for the unreleased in-house imports (``util``, ``magic``, ``ttnsVector``);
* ``tol=`` -> ``rtol=`` keyword shim for SciPy >= 1.14 (numpyVector.py:161,163);
* ``writeOut=False, saveTNSsEachIteration=False`` (HEAD's defaults crash for ndarray
  backends, inexact_Lanczos.py:384-393).

Inputs are regenerated from seeds by ``eigensolvers_amd.generators`` (so they are not
stored); stored are the reference's OUTPUTS: eigenvalues, iteration counts, residuals,
operator products, Gram-Schmidt results, Gram matrices and inner-solve results.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def import_reference():
    sys.path.insert(0, "/root/reference")
    u = types.ModuleType("util")
    u.au2unit = lambda a, unit: a
    u.unit2au = lambda a, unit: a
    sys.modules["util"] = u
    m = types.ModuleType("magic")
    m.ipsh = lambda *a, **k: None
    sys.modules["magic"] = m
    t = types.ModuleType("ttnsVector")
    t.TTNSVector = type("TTNSVector", (), {})
    sys.modules["ttnsVector"] = t
    import scipy.sparse.linalg as spl
    g0, m0 = spl.gcrotmk, spl.minres

    def gcrotmk(A, b, x0=None, tol=None, **kw):
        if tol is not None:
            kw["rtol"] = tol
        return g0(A, b, x0, **kw)

    def minres(A, b, x0=None, tol=None, **kw):
        if tol is not None:
            kw["rtol"] = tol
        return m0(A, b, x0, **kw)

    spl.gcrotmk, spl.minres = gcrotmk, minres
    import inexact_Lanczos
    import numpyVector
    import util_funcs
    return inexact_Lanczos, numpyVector.NumpyVector, util_funcs


def main():
    import scipy.linalg as la
    from eigensolvers_amd.generators import dense_test_matrix, gapped_csr_host, guess_vector
    iL, NumpyVector, uf = import_reference()
    run = lambda *a, **k: iL.inexactLanczosDiagonalization(*a, writeOut=False, saveTNSsEachIteration=False, **k)

    # 1. unittests/test_lanczos.py:14-41 (dense n=100, gcrotmk)
    A, ev = dense_test_matrix(100, 1212)
    y0 = np.random.random(100)                       # continues the seeded legacy stream, as the test does
    opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-4}}
    e, Y, st = run(A, NumpyVector(y0.copy(), opt), 30, 6, 4, 1e-6)
    np.savez(os.path.join(HERE, "lanczos_n100_seed1212.npz"), guess=y0, ev=e,
             vec0=Y[0].array, cumIter=st["cumIter"], residual=st["residual"],
             isConverged=st["isConverged"], exact=ev)

    # 2. unittests/test_lanczosBlock.py:14-41 (3-fold degenerate level, block of 3)
    n, nB, iB = 100, 3, 5
    evb = np.linspace(1, 200, n)
    evb[iB:iB + nB] = evb[iB]
    Ab, _ = dense_test_matrix(n, 1212, evb)
    Ys = la.qr(np.random.rand(n, nB), mode="economic")[0]
    opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-4}}
    e, Y, st = run(Ab, [NumpyVector(Ys[:, i].copy(), opt) for i in range(nB)], evb[iB] + nB / 2, 6, 4, 1e-6)
    np.savez(os.path.join(HERE, "block3_degenerate.npz"), guess=Ys, ev=e, cumIter=st["cumIter"],
             residual=st["residual"], isConverged=st["isConverged"], exact=evb)

    # 3. gapped random-sparse CSR, N=4000, the two solvers of the reference
    N = 4000
    H = gapped_csr_host(N, 32, seed=7)
    g = guess_vector(N, 1)
    for solver in ("minres", "gcrotmk"):
        opt = {"linearSystemArgs": {"linearSolver": solver, "linearIter": 2000, "linear_tol": 1e-10,
                                    "linear_atol": 1e-12}}
        e, Y, st = run(H, NumpyVector(g.copy(), opt), 0.02, 8, 10, 1e-13)
        np.savez(os.path.join(HERE, f"gapped_csr_n4000_{solver}.npz"), ev=e, vec0=Y[0].array,
                 cumIter=st["cumIter"], residual=st["residual"], isConverged=st["isConverged"])

    # 3b. block Lanczos on the same operator: a converging block of 3 (inexact solves), and a
    #     block of 4 that ends in the Gram-Schmidt lindep exit (NaN eigenvalues, :356-359)
    for nb, L_, maxit_, tol_, econv_, tag in ((3, 3, 12, 1e-8, 1e-7, "block3"), (4, 3, 12, 1e-10, 1e-7, "block4_lindep")):
        Qb = la.qr(np.random.default_rng(5).standard_normal((N, nb)), mode="economic")[0]
        opt = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": tol_}}
        import warnings as _w
        with _w.catch_warnings():
            _w.simplefilter("ignore")
            e, Y, st = run(H, [NumpyVector(Qb[:, i].copy(), opt) for i in range(nb)], 0.02, L_, maxit_, econv_)
        np.savez(os.path.join(HERE, f"gapped_csr_n4000_{tag}.npz"), ev=e, cumIter=st["cumIter"],
                 residual=st["residual"], isConverged=st["isConverged"], nvec=len(Y))

    # 4. operator application and the shifted LinearOperator of solve() (numpyVector.py:98-100,152)
    x = np.random.default_rng(11).standard_normal(N)
    v = NumpyVector(x.copy(), {})
    np.savez(os.path.join(HERE, "spmv_n4000.npz"), y=v.applyOp(H).array, yshift=0.02 * x - H @ x)

    # 5. orthogonalize_against_set (numpyVector.py:121-145) incl. the lindep -> None exit
    rng = np.random.default_rng(21)
    Yq = la.qr(rng.standard_normal((N, 7)), mode="economic")[0]
    qs = [NumpyVector(Yq[:, i].copy(), {}) for i in range(7)]
    xv = NumpyVector(rng.standard_normal(N), {})
    out = NumpyVector.orthogonalize_against_set(xv, qs)
    dep = NumpyVector.orthogonalize_against_set(NumpyVector(Yq[:, :3] @ np.array([0.3, -0.2, 0.9]), {}), qs)
    np.savez(os.path.join(HERE, "mgs_step.npz"), out=out.array, dep_is_none=dep is None)

    # 6. Gram builders and their one-vector extensions (numpyVector.py:180-238)
    vecs = [NumpyVector(rng.standard_normal(N), {}) for _ in range(5)]
    S4 = NumpyVector.overlapMatrix(vecs[:4])
    H4 = NumpyVector.matrixRepresentation(H, vecs[:4])
    np.savez(os.path.join(HERE, "gram_n4000.npz"), S=NumpyVector.overlapMatrix(vecs),
             Hm=NumpyVector.matrixRepresentation(H, vecs),
             Sext=NumpyVector.extendOverlapMatrix(vecs, S4),
             Hext=NumpyVector.extendMatrixRepresentation(H, vecs, H4),
             lincomb=NumpyVector.linearCombination(vecs, [0.5, -1.25, 2.0, 0.125, -3.0]).array)

    # 7. the inner solve itself (numpyVector.py:147-178, minres branch) + non-convergence raising
    b = NumpyVector(g / np.linalg.norm(g), {"linearSystemArgs": {"linearSolver": "minres",
                    "linearIter": 2000, "linear_tol": 1e-10}})
    w = NumpyVector.solve(H, b, 0.02)
    import warnings
    raised = False
    try:
        with warnings.catch_warnings():
            NumpyVector.solve(H, NumpyVector(b.array.copy(), {"linearSystemArgs": {
                "linearSolver": "minres", "linearIter": 5, "linear_tol": 1e-12}}), 0.02)
    except UserWarning:
        raised = True
    np.savez(os.path.join(HERE, "solve_n4000_minres.npz"), w=w.array, wnorm=np.linalg.norm(w.array),
             nonconverged_raises=raised)

    # 8. host subspace helpers (util_funcs.py:233-247, 249-289, 360-385)
    Sm = NumpyVector.overlapMatrix(vecs)
    stt = {}
    stt, uS = uf.lowdinOrthoMatrix(Sm, stt)
    evs, uvs = uf.diagonalizeHamiltonian(uS, NumpyVector.matrixRepresentation(H, vecs))
    np.savez(os.path.join(HERE, "subspace_helpers.npz"), uS=uS, evs=evs,
             resid=uf.eigenvalueResidual(np.array([1.0, 2.0, 3.5]), np.array([1.1, 1.9, 3.0])))
    # 9. FEAST (feast.py:126-244) on the reference's own test problem, unittests/test_feast.py:15-43
    import feast as ref_feast
    n, m0 = 100, 6
    evf = np.linspace(1, 200, n)
    np.random.seed(10)
    Qf = la.qr(np.random.rand(n, n))[0]
    Af = Qf.T @ np.diag(evf) @ Qf
    Y0 = np.random.random((n, m0))
    for i in range(m0):
        Y0[:, i] = np.ones(n) * (i + 1)
    Y1 = la.qr(Y0, mode="economic")[0]
    opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-2}}
    Yv = [NumpyVector(Y1[:, i].copy(), opt) for i in range(m0)]
    import warnings as _w2
    with _w2.catch_warnings():
        _w2.simplefilter("ignore")
        evF, YF, stF = ref_feast.feastDiagonalization(Af, Yv, 8, "legendre", 160.0, 166.0, 1e-10, 20, writeOut=False)
    np.savez(os.path.join(HERE, "feast_n100.npz"), A=Af, guess=Y1, ev=evF, outerIter=stF["outerIter"],
             residual=stF["residual"], nvec=len(YF))
    # one quadrature term and the node/weight helper
    gk, wk = uf.quadraturePointsWeights(8, "legendre", positiveHalf=True)
    theta = -(np.pi * 0.5) * (gk[0] - 1)
    z = 163.0 + 3.0 * (np.cos(theta) + 1.0j * np.sin(theta))
    term = ref_feast.calculateQuadrature(Af, NumpyVector(Y1[:, 0].copy(), {"linearSystemArgs": {
        "linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-10, "linear_atol": 1e-12}}), z, 3.0, theta, wk[0], 1.0)
    gt, wt = uf.quadraturePointsWeights(6, "trapezoidal", positiveHalf=False)
    np.savez(os.path.join(HERE, "feast_pieces.npz"), gk=gk, wk=wk, term=term.array, z=z, theta=theta, gt=gt, wt=wt)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()

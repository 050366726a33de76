"""Exploration of the full-size BASELINE configurations on the GPU (parameters for the -m gpu tests).
    python tools/experiments/fullsize_probe.py feast4000 | block8 [N] | single [N] | feast N m0 tol econv maxit"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.linalg as la
import eigensolvers_amd as ea
from eigensolvers_amd.generators import guess_vector

what = sys.argv[1]
ctx = ea.HipContext.default()
if what == "feast4000":
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "../../tests/golden/feast_gapped_n4000.npz"))
    from eigensolvers_amd.generators import gapped_csr_host
    H = ea.HipCsrOperator.from_scipy(gapped_csr_host(4000, 32, seed=7))
    m0 = int(g["m0"])
    Q = la.qr(np.random.default_rng(int(g["seed"])).standard_normal((4000, m0)), mode="economic")[0]
    tol = float(g["linear_tol"])
    opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 2000, "linear_tol": tol, "linear_atol": tol * 1e-2}}
    t = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = ea.feastDiagonalization(H, [ea.HipVector(Q[:, i].copy(), opt) for i in range(m0)], int(g["nc"]), "legendre",
                                            float(g["eMin"]), float(g["eMax"]), float(g["eConv"]), int(g["maxit"]), writeOut=False)
    print("feast4000: outer", st["outerIter"], "golden", int(g["outerIter"]), "res", st["residual"], float(g["residual"]), "t", time.time() - t)
    print(np.max(np.abs(np.sort(ev) - np.sort(g["ev"])) / np.abs(np.sort(g["ev"]))))
elif what in ("block8", "single"):
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    nnz = 32 if N <= 2_000_000 else 64
    H = ea.HipCsrOperator.generate(N, nnz, seed=7)
    for (L, maxit, tol, econv) in ([(5, 6, 1e-6, 1e-8), (10, 2, 1e-10, 1e-11), (9, 2, 1e-9, 1e-10), (12, 2, 1e-11, 1e-12)] if what == "block8" else [(8, 4, 1e-10, 1e-12)]):
        opt = lambda: {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 4000, "linear_tol": tol}}
        if what == "block8":
            Q = la.qr(np.random.default_rng(5).standard_normal((N, 8)), mode="economic")[0]
            v0 = [ea.HipVector(Q[:, i].copy(), opt()) for i in range(8)]
        else:
            v0 = ea.HipVector(guess_vector(N, 1).copy(), opt())
        t = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev, Y, st = ea.inexactLanczosDiagonalization(H, v0, 0.02, L, maxit, econv, writeOut=False)
        dt = time.time() - t
        nb = 8 if what == "block8" else 1
        print(what, N, (L, maxit, tol, econv), "cumIter", st["cumIter"], "conv", st["isConverged"], "res", st["residual"], "t %.1f" % dt)
        if not np.all(np.isnan(ev)):
            res = ea.true_residual_norms(H, ev, Y, nb)
            print("  ev", np.array2string(np.sort(ev[:nb]), precision=13), "\n  resid", res)
            S = ea.HipVector.overlapMatrix(Y[:nb]); print("  |S-I|", np.abs(S - np.eye(nb)).max())
elif what == "feast":
    N = int(sys.argv[2]); m0 = int(sys.argv[3]); tol = float(sys.argv[4]); econv = float(sys.argv[5]); maxit = int(sys.argv[6])
    nnz = 32 if N <= 2_000_000 else 64
    H = ea.HipCsrOperator.generate(N, nnz, seed=7)
    Q = la.qr(np.random.default_rng(9).standard_normal((N, m0)), mode="economic")[0]
    opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 4000, "linear_tol": tol, "linear_atol": tol * 1e-2}}
    t = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = ea.feastDiagonalization(H, [ea.HipVector(Q[:, i].copy(), opt) for i in range(m0)], 16, "legendre", -0.21, 0.21, econv, maxit, writeOut=False)
    inw = np.sort(ev[(ev > -0.21) & (ev < 0.21)])
    print("feast", N, m0, tol, econv, "outer", st["outerIter"], "res", st["residual"], "nin", len(inw), "nvec", len(Y), "t %.1f" % (time.time() - t))
    print(np.array2string(inw, precision=12))
    print("true residuals", ea.true_residual_norms(H, ev, Y, len(Y)))

"""Where the launch-bound FEAST run at N = 4000 (tests/golden/feast_gapped_n4000.npz's case, first argv[1] iterations)
spends its host time: cProfile top entries by own time."""
import cProfile, io, os, pstats, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as la
import eigensolvers_amd as ea
maxit = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, m0 = 4000, 16
H = ea.HipCsrOperator.generate(N, 32, seed=7)
Q = la.qr(np.random.default_rng(9).standard_normal((N, m0)), mode="economic")[0]
opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 2000, "linear_tol": 1e-6, "linear_atol": 1e-8}}
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ev, Y, st = ea.feastDiagonalization(H, [ea.HipVector(Q[:, i].copy(), dict(opt)) for i in range(m0)], 16, "legendre", -0.21, 0.21,
                                        1e-6, maxit, writeOut=False)
pr.disable()
print("FEAST N=4000, %d iteration(s): %.2f s" % (maxit, time.perf_counter() - t))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])

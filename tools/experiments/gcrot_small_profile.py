"""Where a small-N GCROT solve spends its time (host Python vs library calls): dense N = 2000, gcrotmk.  cProfile top entries."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
from eigensolvers_amd.generators import dense_test_matrix
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
A, exact = dense_test_matrix(N, 1212)
H = ea.HipCsrOperator.from_dense(A)
opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-8}}
b = ea.HipVector(np.random.default_rng(1).standard_normal(N), opt)
b.normalize()
sigma = float(exact[N // 2]) + 0.37 * float(exact[N // 2 + 1] - exact[N // 2])
ea.HipVector.solve(H, b, sigma)                     # warm-up (kernel loading)
t = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
w = ea.HipVector.solve(H, b, sigma)
pr.disable()
dt = time.perf_counter() - t
print("solve: %.3f s, stats %s" % (dt, w.last_solve_stats))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])

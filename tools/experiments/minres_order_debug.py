import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eigensolvers_amd as ea
from eigensolvers_amd.generators import gapped_csr_host, guess_vector
Hh = gapped_csr_host(4000, 32, seed=7); guess = guess_vector(4000, 1)
b = guess / np.linalg.norm(guess)
def solve(tag, rtol=1e-6):
    H = ea.HipCsrOperator.from_scipy(Hh)
    B = ea.HipVector(b.copy(), {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": rtol}})
    W = ea.HipVector.solve(H, B, 0.02)
    w = W.array
    print(tag, "its", W.last_solve_stats["iterations"], "w[1875]=%.17g" % w[1875], "norm=%.17g" % np.linalg.norm(w), flush=True)
    return w
w0 = solve("fresh      ")
w1 = solve("again      ")
solve("rtol 1e-10 ", 1e-10)
w2 = solve("after 1e-10")
N = 300000
H2 = ea.HipCsrOperator.generate(N, 32, seed=5)
b2 = ea.HipVector(np.random.default_rng(5).standard_normal(N), {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}})
b2.normalize()
ea.HipVector.solve(H2, b2, 0.02)
w3 = solve("after big  ")
os.environ["HIPEIG_GRAPH"] = "0"
print("max diffs", np.max(np.abs(w1 - w0)), np.max(np.abs(w2 - w0)), np.max(np.abs(w3 - w0)))

"""GCROT and FEAST at N = 1e6 on the GPU: a scale sanity run (no CPU comparator at this size)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as la, eigensolvers_amd as ea
from eigensolvers_amd.generators import guess_vector
N = 1_000_000
H = ea.HipCsrOperator.generate(N, 32, seed=7)
opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 500, "linear_tol": 1e-8, "linear_atol": 1e-14}}
t = time.time()
ev, Y, st = ea.inexactLanczosDiagonalization(H, ea.HipVector(guess_vector(N, 1).copy(), opt), 0.02, 8, 4, 1e-10, writeOut=False)
print("lanczos+gcrotmk N=1e6:", ev[0], st["cumIter"], st["isConverged"], "matvecs last solve", Y[0].last_solve_stats, round(time.time() - t, 2), "s",
      "true residual", ea.true_residual_norms(H, ev, Y, 1)[0])
opt2 = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 3000, "linear_tol": 1e-10}}
ev2, Y2, st2 = ea.inexactLanczosDiagonalization(H, ea.HipVector(guess_vector(N, 1).copy(), opt2), 0.02, 8, 4, 1e-10, writeOut=False)
print("lanczos+minres  N=1e6:", ev2[0], st2["cumIter"], "rel diff", abs(ev[0] - ev2[0]) / abs(ev2[0]))

"""One MINRES solve at N = 2e5 (the size where the captured chunk helps most) - used by graph_rocprof.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
N = 200_000
H = ea.HipCsrOperator.generate(N, 32, seed=7)
b = ea.HipVector(np.random.default_rng(1).standard_normal(N), {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}})
w = ea.HipVector.solve(H, b, 0.02)
print("graph", os.environ.get("HIPEIG_GRAPH", "0"), "iterations", w.last_solve_stats["iterations"], "istop", w.last_solve_stats["istop"], "|w|", w.norm())

#!/usr/bin/env python3
"""Where does a FEAST iteration spend its time?  cProfile of one iteration (host view) at FEAST_N."""
import cProfile
import os
import pstats
import sys
import warnings

import numpy as np
import scipy.linalg as la

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eigensolvers_amd as ea  # noqa: E402

N, m0 = int(os.environ.get("FEAST_N", 200_000)), int(os.environ.get("FEAST_M0", 4))
H = ea.HipCsrOperator.generate(N, 32, seed=7)
Y0 = la.qr(np.random.default_rng(9).standard_normal((N, m0)), mode="economic")[0]
opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 2000, "linear_tol": 1e-4, "linear_atol": 1e-12}}
Y = [ea.HipVector(Y0[:, i].copy(), opt) for i in range(m0)]
warnings.simplefilter("ignore")
pr = cProfile.Profile()
pr.enable()
ev, Yf, st = ea.feastDiagonalization(H, Y, 8, "legendre", -0.21, 0.21, 1e-9, 1, writeOut=False)
ea.HipContext.default().synchronize()
pr.disable()
ps = pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative")
ps.print_stats(35)
ps.sort_stats("tottime").print_stats(20)
print("last solve stats:", Yf[0].last_solve_stats)

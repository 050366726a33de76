#!/usr/bin/env python3
"""The reference's examples/driver_numpyVector.py restated for the MI355X backend: a dense random
symmetric matrix with known spectrum, one guess vector, shift-and-invert Lanczos towards sigma.
The only changes are the backend class, the operator wrapper and the inner solver name (MINRES is
the device solver; the reference example uses gcrotmk)."""
import os
import sys

import numpy as np
import scipy.linalg as la

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigensolvers_amd as ea  # noqa: E402

larger = "--larger" in sys.argv
n, top, target, maxit, L, eConv = (2500, 1400, 1290, 20, 50, 1e-12) if larger else (100, 300, 30, 4, 6, 1e-8)
ev = np.linspace(1, top, n)
np.random.seed(10)
Q = la.qr(np.random.rand(n, n))[0]
A = Q.T @ np.diag(ev) @ Q
options = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 5000, "linear_tol": 1e-4}}
H = ea.HipCsrOperator.from_dense(A)
Y0 = ea.HipVector(np.random.random(n), options)
lf, xf, status = ea.inexactLanczosDiagonalization(H, Y0, target, L, maxit, eConv, writeOut=False)
print({k: status[k] for k in ("cumIter", "isConverged", "residual", "runTime")})
print("Eigenvalue nearest to sigma       ::", round(ea.find_nearest(lf, target)[1], 8))
print("Actual eigenvalue nearest to sigma::", round(ea.find_nearest(ev, target)[1], 8))
print("true residual norm                ::", ea.true_residual_norms(H, lf, xf, 1)[0])

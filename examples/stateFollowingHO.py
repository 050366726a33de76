#!/usr/bin/env python3
"""State following by maximum overlap (reference: examples/stateFollowingHO.py): 1-D harmonic
oscillator in a sinc-DVR, target the state ABOVE the one closest to sigma by picking the Ritz
vector with the largest overlap to a reference vector."""
import os
import sys

import numpy as np
import scipy.linalg as la

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eigensolvers_amd as ea  # noqa: E402
from eigensolvers_amd.generators import sinc_dvr_harmonic  # noqa: E402

Hd, x = sinc_dvr_harmonic(45, (-10, 10))
evEigh, uvEigh = la.eigh(Hd)
options = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 30000, "linear_tol": 1e-4}}
sigma = 11.1
idx = ea.find_nearest(evEigh, sigma)[0]
ref = ea.HipVector(uvEigh[:, idx + 1].copy(), options)
np.random.seed(13)
Y0 = ea.HipVector(np.random.random(45), options)
ev, Y, status = ea.inexactLanczosDiagonalization(ea.HipCsrOperator.from_dense(Hd), Y0, sigma, 5, 200, 1e-10,
                                                 pick=ea.get_pick_function_maxOvlp(ref), writeOut=False)
print("followed state:", ev[0], "reference:", evEigh[idx + 1], "converged:", status["isConverged"],
      "cumIter:", status["cumIter"])

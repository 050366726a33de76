"""Oracle (test infrastructure): CPU restatement of the reference's FEAST driver.

Follows feast.py:45-244 and the quadrature helper util_funcs.py:14-27,146-166, written against
``oracle.numpy_vector.RefVector`` (complex shifts go to SciPy ``gcrotmk`` / ``spsolve`` exactly as
``NumpyVector.solve`` does).  Pinned by ``tests/golden/feast_*.npz`` (outputs of the real
reference) and by the reference's Fortran known-answer file ``data_fortranCode.out``.
Never imported by ``eigensolvers_amd``.
"""
import math
import warnings

import numpy as np
from scipy import special

from .lanczos_ref import basis_transformation, diagonalize_hamiltonian, lowdin_ortho_matrix


def quadrature(nc, quad, positive_half=True):                  # util_funcs.py:146-166
    if quad == "legendre":
        gk, wk = special.roots_legendre(nc)
    elif quad == "hermite":
        gk, wk = special.roots_hermite(nc)
    else:                                                      # "trapezoidal", util_funcs.py:14-27
        dx = 2.0 / nc
        gk = np.array([-1.0 + dx * (i - 1) for i in range(nc)])
        wk = np.full(nc, 2.0 / (nc + 1))
    if positive_half:
        m = gk > 0.0
        gk, wk = gk[m], wk[m]
    return gk, wk


def quadrature_term(A, b, z, radius, angle, weight, ellipse):   # feast.py:45-103
    cls = type(b)
    if abs(z.imag) < 1e-15:
        z, op = z.real, "her"
    else:
        op = "gen"
    Qe = cls.solve(A, b, z, opType=op)
    mult = -0.50 * weight * radius * (ellipse * math.cos(angle) + math.sin(angle) * 1j)
    return cls.real(mult * Qe)


def window_residual(ev, ref, lo, hi):                          # util_funcs.py:249-289 with a range
    idx = [i for i, r in enumerate(ref) if lo <= r <= hi]
    if idx:
        ref, ev = ref[idx], ev[idx]
    return sum(abs(r - e) for r, e in zip(ref, ev)) / sum(abs(e) for e in ev)


def feast(A, Y, nc, quad, eMin, eMax, eConv, maxit, ellipse=1.0, history=None):
    cls = type(Y[0])
    nsub = len(Y)
    radius = (eMax - eMin) * 0.5
    gk, wk = quadrature(nc, quad, True)
    status = {"outerIter": 0, "residual": None, "isConverged": False}
    ev = ref = None
    for it in range(maxit):
        status["outerIter"] = it
        Q = [None] * nsub
        for k in range(len(gk)):
            theta = -(math.pi * 0.5) * (gk[k] - 1)
            z = (eMin + eMax) * 0.5 + radius * (math.cos(theta) + ellipse * 1.0j * math.sin(theta))
            for i in range(nsub):
                t = quadrature_term(A, Y[i], z, radius, theta, wk[k], ellipse)
                Q[i] = t if k == 0 else cls.linearCombination([Q[i], t], [1.0, 1.0])
        S = cls.overlapMatrix(Q)
        Hm = cls.matrixRepresentation(A, Q)
        status, uS = lowdin_ortho_matrix(S, status)
        ev, uv = diagonalize_hamiltonian(uS, Hm)
        Y = basis_transformation(Q, uS @ uv)
        if history is not None:
            history.append(ev.copy())
        if it != 0:
            if len(ref) > len(ev):
                ref = ref[np.argmin(np.abs(ref[:, None] - ev[None, :]), axis=0)]
            elif len(ref) < len(ev):
                raise RuntimeError("Enlarged space?")
            status["residual"] = window_residual(ev, ref, eMin, eMax)
            if status["residual"] < eConv:
                break
        if nsub != len(Y):
            warnings.warn(f"Alert! Got {nsub - len(Y)} dependent vectors")
        nsub = len(Y)
        ref = ev
    return ev, Y, status

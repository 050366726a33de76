"""FEAST contour-integration eigensolver on the ``AbstractVector`` surface.

Drop-in for the reference's ``feastDiagonalization`` (feast.py:126-244; Polizzi, PRB 79, 115112):
all eigenpairs of a Hermitian operator inside [eMin, eMax] from a subspace that is filtered by a
Gauss quadrature of the resolvent along a half contour, one shifted linear solve
``(z_k I - A) q = y`` per (contour point, subspace vector), followed by a Rayleigh-Ritz step in
the Loewdin-orthonormalised filtered space.  Backend-agnostic like the Lanczos driver: only
``solve`` (with a complex shift), scalar multiplication, ``real``, ``linearCombination``,
``overlapMatrix`` and ``matrixRepresentation`` of ``type(Y[0])`` are used.

Multi-GPU ("replicas"): with ``contourComm`` the contour points are dealt round
robin to the ranks, every rank holding the whole operator and whole vectors; the only exchange is
one SUM all-reduce per filtered vector after the quadrature loop (in place of the serial
accumulation of ``updateQ``).  All ranks then carry out the same Rayleigh-Ritz step on identical data.

Quadrature: only nodes with positive abscissa are kept (``positiveHalf``), so ``nc`` nodes mean
``nc/2`` solves per vector (util_funcs.py:146-166).  The reference's ``trapezoidal`` rule is
restated with its quirks (off-by-one abscissae, weights (b-a)/(nc+1); util_funcs.py:14-27).
"""
import math
import time
import warnings

import numpy as np
from scipy import special

from .subspace import basisTransformation, loewdin_transform, ritz_pairs

__all__ = ["feastDiagonalization", "quadraturePointsWeights", "calculateQuadrature", "updateQ",
           "select_within_range", "contour_point"]


def _trapezoidal(nc):                                         # util_funcs.py:14-27, quirks kept
    a, b = -1.0, 1.0
    dx = (b - a) / nc
    pts = np.array([a + dx * (i - 1) for i in range(nc)])
    return pts, np.full(nc, (b - a) / (nc + 1))


def quadraturePointsWeights(nc, quad, positiveHalf=True):
    """Nodes and weights on [-1, 1] (util_funcs.py:146-166)."""
    if quad == "legendre":
        gk, wk = special.roots_legendre(nc)
    elif quad == "hermite":
        gk, wk = special.roots_hermite(nc)
    elif quad == "trapezoidal":
        gk, wk = _trapezoidal(nc)
    else:
        raise ValueError(f"unknown quadrature {quad!r}")
    if positiveHalf:
        keep = gk > 0.0
        gk, wk = gk[keep], wk[keep]
    return gk, wk


def select_within_range(values, lo, hi):
    """(values inside [lo, hi], their indices) - util_funcs.py:112-125."""
    idx = [i for i, v in enumerate(values) if lo <= v <= hi]
    return np.array([values[i] for i in idx]), idx


def _eigenvalue_change_in_window(ev, reference, lo, hi):
    """util_funcs.py:249-289 with an eigenvalue range: compare only pairs whose REFERENCE value
    lies in the window (all pairs when none does)."""
    if lo > hi:
        warnings.warn("emin is greater than emax. Moving forward with swapped values")
        lo, hi = hi, lo
    idx = select_within_range(reference, lo, hi)[1]
    if len(idx) >= 1:
        reference, ev = reference[idx], ev[idx]
    num = sum(abs(r - e) for r, e in zip(reference, ev))
    den = sum(abs(e) for e in ev)
    return num / den


def contour_point(eMin, eMax, g, contourEllipseFactor=1.0):
    """Angle and complex node for abscissa g (feast.py:191-195, Polizzi eq. 13)."""
    theta = -(math.pi * 0.5) * (g - 1)
    radius = (eMax - eMin) * 0.5
    z = (eMin + eMax) * 0.5 + radius * (math.cos(theta) + contourEllipseFactor * 1.0j * math.sin(theta))
    return theta, z


def calculateQuadrature(Amat, guess_b, z, radius, angle, weight, contourEllipseFactor):
    """One quadrature term -0.5 w r Re{ e^{i theta} (z - A)^{-1} b } (feast.py:45-103)."""
    b = guess_b
    cls = b.__class__
    if abs(z.imag) < 1e-15:
        opType, z = "her", z.real
    else:
        opType = "gen"
    phase = contourEllipseFactor * math.cos(angle) + math.sin(angle) * 1j
    if b.hasExactAddition:
        Qe = cls.solve(Amat, b, z, opType=opType)
        return cls.real((-0.50 * weight * radius * phase) * Qe)
    mult = -0.25 * weight * radius                             # Polizzi (12): both half planes
    p1 = cls.solve(Amat, b, z, opType=opType)
    p2 = cls.solve(Amat, b, z.conjugate(), opType=opType)
    return cls.linearCombination([p1, p2], [mult * phase, mult * phase.conjugate()])


def _quadrature_terms_block(cls, Amat, guesses, z, radius, angle, weight, contourEllipseFactor):
    """The quadrature terms of ONE contour point for all guesses through the backend's block solve hook, when it has one
    (``HipVector.solveBlock``: the solves share operator and shift, feast.py:198-200, so they can advance in lock step on
    block products).  Same arithmetic per vector as ``calculateQuadrature``; ``None`` = take the one-by-one path."""
    b0 = guesses[0]
    if (not hasattr(cls, "solveBlock") or not b0.hasExactAddition or len(guesses) < 2
            or not getattr(b0, "options", {}).get("blockSolve", True)):
        return None
    if abs(z.imag) < 1e-15:
        opType, z = "her", z.real
    else:
        opType = "gen"
    phase = contourEllipseFactor * math.cos(angle) + math.sin(angle) * 1j
    sols = cls.solveBlock(Amat, list(guesses), z, opType=opType)
    return [cls.real((-0.50 * weight * radius * phase) * Qe) for Qe in sols]


def updateQ(Q, im0, Qquad_k, k):
    """Accumulate the k-th quadrature term into the im0-th filtered vector (feast.py:105-121)."""
    if k == 0:
        Q[im0] = Qquad_k
    else:
        Q[im0] = Qquad_k.__class__.linearCombination([Q[im0], Qquad_k], [1.0, 1.0])
    return Q


def feastDiagonalization(A, Y, nc, quad, eMin, eMax, eConv, maxit, contourEllipseFactor=1.0,
                         writeOut=True, eShift=0.0, convertUnit="au", outFileName=None,
                         summaryFileName=None, contourComm=None):
    """Arguments and returns as the reference (feast.py:126-165): ``(ev, Y, status)``.

    ``contourComm`` (not in the reference): an object with ``rank``, ``nranks`` and
    ``allreduce(vector) -> vector`` (e.g. ``distributed.ContourReplicas``); contour point k is then
    solved on rank ``k % nranks`` only."""
    if convertUnit != "au":
        raise NotImplementedError("unit conversion needs the reference's in-house `util` module")
    cls = type(Y[0])
    nsub = len(Y)
    assert eMax > eMin
    radius = (eMax - eMin) * 0.5
    gk, wk = quadraturePointsWeights(nc, quad, positiveHalf=True)
    status = {"flagAddition": Y[0].hasExactAddition, "outerIter": 0, "quadrature": 0, "isConverged": False,
              "phase": 1, "residual": None, "startTime": time.time(), "runTime": 0.0, "converged": False}
    summary = open(summaryFileName or "summary_feast.out", "w") if writeOut else None
    if summary:
        summary.write("startingPoint\n")
    ev, ref_ev = None, None
    for it in range(maxit):
        status["outerIter"] = it
        Q = [None] * nsub
        for k in range(len(gk)):
            if contourComm is not None and k % contourComm.nranks != contourComm.rank:
                continue
            status["quadrature"] = k
            theta, z = contour_point(eMin, eMax, gk[k], contourEllipseFactor)
            terms = _quadrature_terms_block(cls, A, Y[:nsub], z, radius, theta, wk[k], contourEllipseFactor)
            for im0 in range(nsub):
                term = terms[im0] if terms is not None else \
                    calculateQuadrature(A, Y[im0], z, radius, theta, wk[k], contourEllipseFactor)
                Q = updateQ(Q, im0, term, 0 if Q[im0] is None else 1)
        if contourComm is not None:
            for im0 in range(nsub):
                mine = Q[im0] if Q[im0] is not None else 0.0 * Y[im0]      # a rank without contour points
                Q[im0] = contourComm.allreduce(mine)
        # Rayleigh-Ritz in the Loewdin-orthonormalised filtered space (feast.py:203-215)
        S = cls.overlapMatrix(Q)
        Hm = cls.matrixRepresentation(A, Q)
        independent, X = loewdin_transform(S)
        status["lindep"] = not independent
        ev, U = ritz_pairs(X, Hm)
        Y = basisTransformation(Q, X @ U)
        if it != 0:
            if len(ref_ev) > len(ev):
                ref_ev = ref_ev[np.argmin(np.abs(ref_ev[:, None] - ev[None, :]), axis=0)]
            elif len(ref_ev) < len(ev):
                raise RuntimeError(f"ref_ev={ref_ev} but ev={ev}. Enlarged space?")
            residual = _eigenvalue_change_in_window(ev, ref_ev, eMin, eMax)
            status["runTime"] = time.time() - status["startTime"]
            status["residual"] = residual
            if summary:
                summary.write("{:>4} ".format(it) + " ".join(f"{e - eShift:.10f}" for e in ev)
                              + f" {residual:5.4e} {status['runTime']:.2f}\n")
                summary.flush()
            if residual < eConv:
                status["converged"] = True        # (the reference's "isConverged" key is never updated; kept so)
                break
        if nsub != len(Y):
            warnings.warn(f"Alert! Got {nsub - len(Y)} dependent vectors")
        nsub = len(Y)
        ref_ev = ev
    if summary:
        summary.write("endingPoint\n")
        summary.close()
    return ev, Y, status

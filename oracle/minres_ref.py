"""Oracle (test infrastructure): NumPy restatement of MINRES as the reference uses it.

The reference's inner solve is ``scipy.sparse.linalg.minres(linOp, b, x0, tol=...,
maxiter=...)`` (numpyVector.py:163) on ``linOp = sigma*x - H@x`` (numpyVector.py:152).
SciPy is a third-party dependency outside the reference tree; this file restates the
Paige-Saunders recurrences in the order SciPy 1.15.3 evaluates them (no
preconditioner, shift = 0, zero initial guess as on the Lanczos path,
inexact_Lanczos.py:99) so the device solver can be compared iteration by iteration.

Returns the same ``(x, info)`` pair plus an optional trace of per-iteration scalars.
``info`` is ``maxiter`` only when the iteration limit was the stopping reason
(istop == 6) and 0 otherwise, which is what makes numpyVector.py:175-177 raise.
"""
import math

import numpy as np


def minres(matvec, b, rtol=1e-5, maxiter=None, trace=None, dot=np.dot):
    """``dot`` may be replaced by a distributed inner product (row-partitioned tests)."""
    b = np.asarray(b, dtype=np.float64)
    n = b.shape[0]
    if maxiter is None:
        maxiter = 5 * n
    eps = np.finfo(np.float64).eps
    x = np.zeros(n)

    r1 = b.copy()
    y = r1
    beta1 = float(dot(r1, y))
    if beta1 == 0.0:
        return x, 0, 0, 0
    if float(dot(b, b)) == 0.0:
        return b.copy(), 0, 0, 0
    beta1 = math.sqrt(beta1)

    oldb = 0.0
    beta = beta1
    dbar = 0.0
    epsln = 0.0
    phibar = beta1
    tnorm2 = 0.0
    gmax = 0.0
    gmin = np.finfo(np.float64).max
    cs = -1.0
    sn = 0.0
    w = np.zeros(n)
    w2 = np.zeros(n)
    r2 = r1
    istop = 0
    itn = 0

    while itn < maxiter:
        itn += 1
        s = 1.0 / beta
        v = s * y
        y = matvec(v)
        if itn >= 2:
            y = y - (beta / oldb) * r1
        alfa = float(dot(v, y))
        y = y - (alfa / beta) * r2
        r1 = r2
        r2 = y
        oldb = beta
        beta = float(dot(r2, y))
        if beta < 0:
            raise ValueError("non-symmetric matrix")
        beta = math.sqrt(beta)
        tnorm2 += alfa * alfa + oldb * oldb + beta * beta
        if itn == 1 and beta / beta1 <= 10 * eps:
            istop = -1

        oldeps = epsln
        delta = cs * dbar + sn * alfa
        gbar = sn * dbar - cs * alfa
        epsln = sn * beta
        dbar = -cs * beta
        root = math.sqrt(gbar * gbar + dbar * dbar)   # np.linalg.norm([gbar, dbar])

        gamma = max(math.sqrt(gbar * gbar + beta * beta), eps)
        cs = gbar / gamma
        sn = beta / gamma
        phi = cs * phibar
        phibar = sn * phibar

        denom = 1.0 / gamma
        w1 = w2
        w2 = w
        w = (v - oldeps * w1 - delta * w2) * denom
        x = x + phi * w

        gmax = max(gmax, gamma)
        gmin = min(gmin, gamma)

        Anorm = math.sqrt(tnorm2)
        ynorm = math.sqrt(float(dot(x, x)))
        epsx = Anorm * ynorm * eps
        rnorm = phibar
        test1 = math.inf if (ynorm == 0 or Anorm == 0) else rnorm / (Anorm * ynorm)
        test2 = math.inf if Anorm == 0 else root / Anorm
        Acond = gmax / gmin

        if istop == 0:
            if 1 + test2 <= 1:
                istop = 2
            if 1 + test1 <= 1:
                istop = 1
            if itn >= maxiter:
                istop = 6
            if Acond >= 0.1 / eps:
                istop = 4
            if epsx >= beta1:
                istop = 3
            if test2 <= rtol:
                istop = 2
            if test1 <= rtol:
                istop = 1
        if trace is not None:
            trace.append(dict(itn=itn, alfa=alfa, beta=beta, rnorm=rnorm, ynorm=ynorm,
                              Anorm=Anorm, test1=test1, test2=test2, istop=istop))
        if istop != 0:
            break

    info = maxiter if istop == 6 else 0
    return x, info, itn, istop

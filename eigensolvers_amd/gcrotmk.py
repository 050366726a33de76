"""Flexible GCROT(m,k) with device-resident vectors (``linearSolver="gcrotmk"``).

The reference's tests and examples all select ``gcrotmk`` (numpyVector.py:161 ->
``scipy.sparse.linalg.gcrotmk``; SciPy is a third-party dependency outside the reference tree).
This module restates the algorithm SciPy 1.15.3 implements - de Sturler's GCROT with the
simplified flexible update of Hicken & Zingg (SIAM J. Sci. Comput. 32, 172 (2010)): an outer
loop that keeps k pairs (c, u) with c = A u orthonormal, and an inner FGMRES(m) Arnoldi
process whose Krylov vectors are orthogonalised first against the c's and then against each
other by modified Gram-Schmidt, the Hessenberg least-squares problem being updated column by
column through a QR insert.  As the reference calls it: no preconditioner, zero initial guess,
empty recycle space, ``truncate='oldest'``, m = k = 20; convergence when
``||b - A x|| <= max(atol, rtol*||b||)``; ``info`` = 0 or the number of outer iterations spent.

Every N-length vector lives on the GPU and every N-length operation is a ``libhipeig`` kernel
(operator sweep, dot, axpy, scale); only scalars and the (m+1) x m Hessenberg algebra are on the
host - the same split as the Lanczos loop itself.  No vector ever crosses PCIe.
"""
import ctypes as C

import numpy as np
from scipy.linalg import lstsq, qr_insert

from . import _lib


class _Ops:
    """Thin helpers over the C ABI for raw device buffers of one length."""

    def __init__(self, ctx, n):
        self.ctx, self.n, self.h = ctx, n, ctx.handle

    def new(self):
        return self.ctx.alloc(self.n)

    def copy(self, src):
        out = self.new()
        _lib.call("hipeig_vec_copy", self.h, out.ptr, src.ptr, self.n)
        return out

    def dot(self, a, b):
        out = C.c_double()
        _lib.call("hipeig_dot", self.h, self.n, a.ptr, b.ptr, C.byref(out))
        return out.value

    def nrm2(self, a):
        out = C.c_double()
        _lib.call("hipeig_nrm2", self.h, self.n, a.ptr, C.byref(out))
        return out.value

    def axpy(self, alpha, x, y):                       # y += alpha * x
        _lib.call("hipeig_axpby", self.h, self.n, float(alpha), x.ptr, 1.0, y.ptr)

    def scal(self, alpha, x):                          # x *= alpha
        _lib.call("hipeig_scale", self.h, self.n, float(alpha), x.ptr, x.ptr)

    def scaled(self, alpha, x):                        # new vector alpha * x
        out = self.new()
        _lib.call("hipeig_scale", self.h, self.n, float(alpha), x.ptr, out.ptr)
        return out


def _fgmres(ops, matvec, v0, m, atol, cs):
    """Inner Arnoldi process: A [v_0..v_j] = C B + V H with H held as Q R.

    Returns (Q, R, B, vs, y, res); without a preconditioner the z vectors are the v's."""
    vs = [v0]
    B = np.zeros((len(cs), m))
    Q = np.ones((1, 1))
    R = np.zeros((1, 0))
    eps = np.finfo(np.float64).eps
    breakdown = False
    j = 0
    for j in range(m):
        w = matvec(vs[-1])
        w_norm = ops.nrm2(w)
        for i, c in enumerate(cs):                     # (1 - C C^T) A : project out the recycle space
            alpha = ops.dot(c, w)
            B[i, j] = alpha
            ops.axpy(-alpha, c, w)
        hcur = np.zeros(j + 2)
        for i, v in enumerate(vs):                     # modified Gram-Schmidt against V
            alpha = ops.dot(v, w)
            hcur[i] = alpha
            ops.axpy(-alpha, v, w)
        hcur[j + 1] = ops.nrm2(w)
        with np.errstate(over="ignore", divide="ignore"):
            alpha = 1 / hcur[-1]
        if np.isfinite(alpha):
            ops.scal(alpha, w)
        if not (hcur[-1] > eps * w_norm):
            breakdown = True                           # w in the span of the previous vectors (or NaN)
        vs.append(w)
        Q2 = np.zeros((j + 2, j + 2), order="F")
        Q2[:j + 1, :j + 1] = Q
        Q2[j + 1, j + 1] = 1
        R2 = np.zeros((j + 2, j), order="F")
        R2[:j + 1, :] = R
        Q, R = qr_insert(Q2, R2, hcur, j, which="col", overwrite_qru=True, check_finite=False)
        res = abs(Q[0, -1])                            # residual of the Hessenberg LSQ problem
        if res < atol or breakdown:
            break
    if not np.isfinite(R[j, j]):
        raise np.linalg.LinAlgError()
    y, _, _, _ = lstsq(R[:j + 1, :j + 1], Q[0, :j + 1].conj())
    return Q, R, B[:, :j + 1], vs, y, res


def gcrotmk_device(ctx, matvec, b, n, rtol=1e-5, atol=0.0, maxiter=1000, m=20, k=None):
    """Solve A x = b; ``matvec(buf) -> new buf`` applies A on the device.

    Returns ``(x_buf, info, stats)`` with SciPy's ``info`` convention."""
    ops = _Ops(ctx, n)
    if k is None:
        k = m
    x = ops.new()
    _lib.call("hipeig_vec_fill", ctx.handle, x.ptr, n, 0.0)
    r = ops.copy(b)
    b_norm = ops.nrm2(b)
    if not np.isfinite(b_norm):
        raise ValueError("RHS must contain only finite numbers")
    atol = max(float(atol), float(rtol) * float(b_norm))
    stats = {"outer": 0, "matvecs": 0}
    if b_norm == 0:
        return ops.copy(b), 0, stats

    def mv(v):
        stats["matvecs"] += 1
        return matvec(v)

    CU = []
    j_outer = -1
    for j_outer in range(maxiter):
        beta = ops.nrm2(r)
        beta_tol = max(atol, rtol * b_norm)
        if beta <= beta_tol and (j_outer > 0 or CU):
            r = mv(x)                                  # recompute the residual: r = b - A x
            ops.scal(-1.0, r)
            ops.axpy(1.0, b, r)
            beta = ops.nrm2(r)
        if beta <= beta_tol:
            j_outer = -1
            break
        ml = m + max(k - len(CU), 0)
        cs = [c for c, u in CU]
        try:
            Q, R, B, vs, y, pres = _fgmres(ops, mv, ops.scaled(1.0 / beta, r), ml,
                                           atol=max(atol, rtol * b_norm) / beta, cs=cs)
            y = y * beta
        except np.linalg.LinAlgError:
            break
        # new outer pair: ux = (Z - U B) y, cx = V H y, normalised so that cx = A ux, |cx| = 1
        ux = ops.scaled(y[0], vs[0])
        for z, yc in zip(vs[1:], y[1:]):
            ops.axpy(yc, z, ux)
        by = B.dot(y)
        for (c, u), byc in zip(CU, by):
            ops.axpy(-byc, u, ux)
        with np.errstate(invalid="ignore"):
            hy = Q.dot(R.dot(y))
        cx = ops.scaled(hy[0], vs[0])
        for v, hyc in zip(vs[1:], hy[1:]):
            ops.axpy(hyc, v, cx)
        try:
            alpha = 1 / ops.nrm2(cx)
            if not np.isfinite(alpha):
                raise FloatingPointError()
        except (FloatingPointError, ZeroDivisionError):
            continue
        ops.scal(alpha, cx)
        ops.scal(alpha, ux)
        gamma = ops.dot(cx, r)
        ops.axpy(-gamma, cx, r)
        ops.axpy(gamma, ux, x)
        while len(CU) >= k and CU:                     # truncate='oldest'
            del CU[0]
        CU.append((cx, ux))
    stats["outer"] = j_outer + 1
    return x, j_outer + 1, stats

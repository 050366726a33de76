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

    def mgs_project(self, vs, w):
        """Sequential MGS of w against vs on the device; returns the coefficients."""
        out = np.empty(len(vs))
        if len(vs):
            tab = (C.c_void_p * len(vs))(*[v.ptr for v in vs])
            _lib.call("hipeig_mgs_project", self.h, self.n, len(vs), C.cast(tab, C.POINTER(C.c_void_p)), w.ptr,
                      out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def arnoldi_step(self, vs, w):
        """||w||, sequential MGS of w against vs, ||w|| again and w /= ||w|| (when finite) in ONE call
        and one host round trip; returns (norm before, coefficients, norm after)."""
        m = len(vs)
        out = np.empty(m + 2)
        tab = (C.c_void_p * max(m, 1))(*[v.ptr for v in vs])
        _lib.call("hipeig_arnoldi_step", self.h, self.n, m, C.cast(tab, C.POINTER(C.c_void_p)), w.ptr,
                  out.ctypes.data_as(C.POINTER(C.c_double)))
        return float(np.sqrt(out[0])), out[1:m + 1], float(np.sqrt(out[m + 1]))

    def combine(self, coeffs, vecs):
        """sum_i coeffs[i] * vecs[i] in one pass."""
        out = self.new()
        cf = np.ascontiguousarray(coeffs, dtype=np.float64)
        tab = (C.c_void_p * len(vecs))(*[v.ptr for v in vecs])
        _lib.call("hipeig_lincomb", self.h, self.n, len(vecs), cf.ctypes.data_as(C.POINTER(C.c_double)),
                  C.cast(tab, C.POINTER(C.c_void_p)), out.ptr)
        return out


class _PairOps:
    """Complex vectors as (re, im) pairs of real device buffers: every complex operation is a
    few real kernels (the conjugated product is two batched real reductions).  Used for the
    complex-shifted solves of the FEAST contour, (z*I - H) x = b with real H."""

    dtype = np.complex128

    def __init__(self, ctx, n):
        self.r = _Ops(ctx, n)

    def new(self):
        return (self.r.new(), self.r.new())

    def zeros(self):
        a = self.new()
        for part in a:
            _lib.call("hipeig_vec_fill", self.r.h, part.ptr, self.r.n, 0.0)
        return a

    def copy(self, a):
        return (self.r.copy(a[0]), self.r.copy(a[1]))

    def _dots2(self, a, x):                            # [a_re . x, a_im . x] in one batched sweep
        out = np.empty(2)
        tab = (C.c_void_p * 2)(a[0].ptr, a[1].ptr)
        _lib.call("hipeig_multi_dot", self.r.h, self.r.n, 2, C.cast(tab, C.POINTER(C.c_void_p)), x.ptr,
                  out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def dot(self, a, b):                               # conj(a) . b, like BLAS zdotc
        p, q = self._dots2(a, b[0]), self._dots2(a, b[1])      # p = [ar.br, ai.br], q = [ar.bi, ai.bi]
        return complex(p[0] + q[1], q[0] - p[1])

    def nrm2(self, a):
        return float(np.sqrt(self.r.dot(a[0], a[0]) + self.r.dot(a[1], a[1])))

    def axpy(self, alpha, x, y):                       # y += alpha * x
        alpha = complex(alpha)
        self.r.axpy(alpha.real, x[0], y[0])
        self.r.axpy(alpha.real, x[1], y[1])
        if alpha.imag != 0.0:
            self.r.axpy(-alpha.imag, x[1], y[0])
            self.r.axpy(alpha.imag, x[0], y[1])

    def scal(self, alpha, x):
        alpha = complex(alpha)
        if alpha.imag == 0.0:
            self.r.scal(alpha.real, x[0])
            self.r.scal(alpha.real, x[1])
        else:
            t = self.scaled(alpha, x)
            _lib.call("hipeig_vec_copy", self.r.h, x[0].ptr, t[0].ptr, self.r.n)
            _lib.call("hipeig_vec_copy", self.r.h, x[1].ptr, t[1].ptr, self.r.n)

    def scaled(self, alpha, x):
        alpha = complex(alpha)
        out = (self.r.scaled(alpha.real, x[0]), self.r.scaled(alpha.real, x[1]))
        if alpha.imag != 0.0:
            self.r.axpy(-alpha.imag, x[1], out[0])
            self.r.axpy(alpha.imag, x[0], out[1])
        return out

    def mgs_project(self, vs, w):
        out = np.empty(2 * len(vs))
        if len(vs):
            tr = (C.c_void_p * len(vs))(*[v[0].ptr for v in vs])
            ti = (C.c_void_p * len(vs))(*[v[1].ptr for v in vs])
            _lib.call("hipeig_pair_mgs_project", self.r.h, self.r.n, len(vs), C.cast(tr, C.POINTER(C.c_void_p)),
                      C.cast(ti, C.POINTER(C.c_void_p)), w[0].ptr, w[1].ptr, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out[0::2] + 1j * out[1::2]

    def arnoldi_step(self, vs, w):
        m = len(vs)
        out = np.empty(2 * m + 2)
        tr = (C.c_void_p * max(m, 1))(*[v[0].ptr for v in vs])
        ti = (C.c_void_p * max(m, 1))(*[v[1].ptr for v in vs])
        _lib.call("hipeig_pair_arnoldi_step", self.r.h, self.r.n, m, C.cast(tr, C.POINTER(C.c_void_p)),
                  C.cast(ti, C.POINTER(C.c_void_p)), w[0].ptr, w[1].ptr, out.ctypes.data_as(C.POINTER(C.c_double)))
        h = out[1:2 * m + 1]
        return float(np.sqrt(out[0])), h[0::2] + 1j * h[1::2], float(np.sqrt(out[2 * m + 1]))

    def combine(self, coeffs, vecs):
        cf = np.asarray(coeffs, dtype=np.complex128)
        parts = [v[0] for v in vecs] + [v[1] for v in vecs]
        re = self.r.combine(np.concatenate([cf.real, -cf.imag]), parts)
        im = self.r.combine(np.concatenate([cf.imag, cf.real]), parts)
        return (re, im)


def _fgmres(ops, matvec, v0, m, atol, cs):
    """Inner Arnoldi process: A [v_0..v_j] = C B + V H with H held as Q R.

    Returns (Q, R, B, vs, y, res); without a preconditioner the z vectors are the v's."""
    dt = getattr(ops, "dtype", np.float64)
    vs = [v0]
    B = np.zeros((len(cs), m), dtype=dt)
    Q = np.ones((1, 1), dtype=dt)
    R = np.zeros((1, 0), dtype=dt)
    eps = np.finfo(np.float64).eps
    breakdown = False
    j = 0
    for j in range(m):
        w = matvec(vs[-1])
        # ||w||; (1 - C C^H) A, then modified Gram-Schmidt against V: one sequential sweep over the
        # columns of [C, V], dot and update per column in that order; ||w|| again and w /= ||w|| when
        # that is finite - all in one device call with the scalars copied back once
        w_norm, coef, h_last = ops.arnoldi_step(list(cs) + vs, w)
        B[:, j] = coef[:len(cs)]
        hcur = np.zeros(j + 2, dtype=dt)
        hcur[:len(vs)] = coef[len(cs):]
        hcur[j + 1] = h_last
        if not (hcur[-1].real > eps * w_norm):
            breakdown = True                           # w in the span of the previous vectors (or NaN)
        vs.append(w)
        Q2 = np.zeros((j + 2, j + 2), dtype=dt, order="F")
        Q2[:j + 1, :j + 1] = Q
        Q2[j + 1, j + 1] = 1
        R2 = np.zeros((j + 2, j), dtype=dt, order="F")
        R2[:j + 1, :] = R
        Q, R = qr_insert(Q2, R2, hcur, j, which="col", overwrite_qru=True, check_finite=False)
        res = abs(Q[0, -1])                            # residual of the Hessenberg LSQ problem
        if res < atol or breakdown:
            break
    if not np.isfinite(R[j, j]):
        raise np.linalg.LinAlgError()
    y, _, _, _ = lstsq(R[:j + 1, :j + 1], Q[0, :j + 1].conj())
    return Q, R, B[:, :j + 1], vs, y, res


def gcrotmk_device(ctx, matvec, b, n, rtol=1e-5, atol=0.0, maxiter=1000, m=20, k=None, complex_pairs=False, x0=None):
    """Solve A x = b; ``matvec(buf) -> new buf`` applies A on the device.

    With ``complex_pairs`` the vectors are (re, im) pairs of device buffers and the arithmetic is
    complex (SciPy's gcrotmk on a complex LinearOperator).  Returns ``(x, info, stats)`` with
    SciPy's ``info`` convention."""
    ops = _PairOps(ctx, n) if complex_pairs else _Ops(ctx, n)
    if k is None:
        k = m
    if x0 is not None:                                 # SciPy: x = x0, r = b - A x0
        x = ops.copy(x0)
        r = matvec(x)
        ops.scal(-1.0, r)
        ops.axpy(1.0, b, r)
    else:
        if complex_pairs:
            x = ops.zeros()
        else:
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
        by = B.dot(y)
        ux = ops.combine(np.concatenate([y, -by]), vs[:len(y)] + [u for c, u in CU])
        with np.errstate(invalid="ignore"):
            hy = Q.dot(R.dot(y))
        cx = ops.combine(hy, vs[:len(hy)])
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

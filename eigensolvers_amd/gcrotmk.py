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

    def __init__(self, ctx, n, cols_per_pass=1):
        self.ctx, self.n, self.h = ctx, n, ctx.handle
        self.cols_per_pass = int(cols_per_pass)        # Arnoldi sweep: 1 = scipy's order of rounding, 4 = blocked (hipeig.h)

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
        _lib.call("hipeig_arnoldi_step_p", self.h, self.n, m, C.cast(tab, C.POINTER(C.c_void_p)), w.ptr,
                  out.ctypes.data_as(C.POINTER(C.c_double)), self.cols_per_pass)
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

    def __init__(self, ctx, n, cols_per_pass=1):
        self.r = _Ops(ctx, n)
        self.cols_per_pass = int(cols_per_pass)

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
        _lib.call("hipeig_pair_arnoldi_step_p", self.r.h, self.r.n, m, C.cast(tr, C.POINTER(C.c_void_p)),
                  C.cast(ti, C.POINTER(C.c_void_p)), w[0].ptr, w[1].ptr, out.ctypes.data_as(C.POINTER(C.c_double)),
                  self.cols_per_pass)
        h = out[1:2 * m + 1]
        return float(np.sqrt(out[0])), h[0::2] + 1j * h[1::2], float(np.sqrt(out[2 * m + 1]))

    def arnoldi_begin(self, vs, w, slot):
        """Enqueue the step without waiting for its scalars (pinned slot ``slot``); ``arnoldi_end`` collects them."""
        m = len(vs)
        tr = (C.c_void_p * max(m, 1))(*[v[0].ptr for v in vs])
        ti = (C.c_void_p * max(m, 1))(*[v[1].ptr for v in vs])
        _lib.call("hipeig_pair_arnoldi_step_begin", self.r.h, self.r.n, m, C.cast(tr, C.POINTER(C.c_void_p)),
                  C.cast(ti, C.POINTER(C.c_void_p)), w[0].ptr, w[1].ptr, self.cols_per_pass, int(slot))

    def arnoldi_end(self, m, slot):
        out = np.empty(2 * m + 2)
        _lib.call("hipeig_arnoldi_step_end", self.r.h, int(slot), 2 * m + 2, out.ctypes.data_as(C.POINTER(C.c_double)))
        h = out[1:2 * m + 1]
        return float(np.sqrt(out[0])), h[0::2] + 1j * h[1::2], float(np.sqrt(out[2 * m + 1]))

    SPLIT_MAX_COLS = 62          # 2m + 2 doubles must fit a pinned slot (hipeig.h)
    BATCH_MAX_N = 8192           # lengths at which a step is ONE workgroup: several steps share a launch

    @staticmethod
    def arnoldi_begin_batch(opss, reqs):
        """The steps ``reqs = [(columns, w), ...]`` of up to 16 right-hand sides in one launch, a workgroup each (slots
        0, 1, ...; collect with ``arnoldi_end``).  Returns False, having done nothing, when the library declines (vectors
        longer than one workgroup handles)."""
        cnt, r = len(reqs), opss[0].r
        ms = (C.c_int * cnt)(*[len(vs) for vs, _ in reqs])
        tr, ti = (C.c_void_p * (64 * cnt))(), (C.c_void_p * (64 * cnt))()
        for i, (vs, _) in enumerate(reqs):
            for j, v in enumerate(vs):
                tr[64 * i + j], ti[64 * i + j] = v[0].ptr, v[1].ptr
        wr = (C.c_void_p * cnt)(*[w[0].ptr for _, w in reqs])
        wi = (C.c_void_p * cnt)(*[w[1].ptr for _, w in reqs])
        PP = C.POINTER(C.c_void_p)
        rc = getattr(_lib.load(), "hipeig_pair_arnoldi_step_batch_begin")(r.h, r.n, cnt, ms, C.cast(tr, PP), C.cast(ti, PP),
                                                                          C.cast(wr, PP), C.cast(wi, PP))
        if rc == 5:
            return False
        _lib.check(rc, "hipeig_pair_arnoldi_step_batch_begin")
        return True

    def combine(self, coeffs, vecs):
        cf = np.asarray(coeffs, dtype=np.complex128)
        parts = [v[0] for v in vecs] + [v[1] for v in vecs]
        re = self.r.combine(np.concatenate([cf.real, -cf.imag]), parts)
        im = self.r.combine(np.concatenate([cf.imag, cf.real]), parts)
        return (re, im)


def _fgmres(ops, v0, m, atol, cs):
    """Inner Arnoldi process: A [v_0..v_j] = C B + V H with H held as Q R.  A generator: every operator application is
    ``w = yield ("mv", v)`` and every orthogonalisation step ``yield ("arn", columns, w)`` - the driver decides whether
    that is one product / one device call, or a column of a block product and one of a batch of steps enqueued back to
    back for the right-hand sides of a lock-step solve.

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
        w = yield ("mv", vs[-1])
        # ||w||; (1 - C C^H) A, then modified Gram-Schmidt against V: one sequential sweep over the
        # columns of [C, V], dot and update per column in that order; ||w|| again and w /= ||w|| when
        # that is finite - all in one device call with the scalars copied back once
        w_norm, coef, h_last = yield ("arn", list(cs) + vs, w)
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


def _gcrotmk(ops, ctx, b, n, rtol, atol, maxiter, m, k, complex_pairs, x0, stats):
    """The solver as a generator (``A v = yield v``); returns ``(x, info)`` with SciPy's ``info`` convention."""
    if k is None:
        k = m

    def mv(v):
        stats["matvecs"] += 1
        return v

    if x0 is not None:                                 # SciPy: x = x0, r = b - A x0
        x = ops.copy(x0)
        r = yield ("mv", x)
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
    if b_norm == 0:
        return ops.copy(b), 0

    CU = []
    j_outer = -1
    for j_outer in range(maxiter):
        beta = ops.nrm2(r)
        beta_tol = max(atol, rtol * b_norm)
        if beta <= beta_tol and (j_outer > 0 or CU):
            r = yield ("mv", mv(x))                    # recompute the residual: r = b - A x
            ops.scal(-1.0, r)
            ops.axpy(1.0, b, r)
            beta = ops.nrm2(r)
        if beta <= beta_tol:
            j_outer = -1
            break
        ml = m + max(k - len(CU), 0)
        cs = [c for c, u in CU]
        try:
            inner = _fgmres(ops, ops.scaled(1.0 / beta, r), ml, atol=max(atol, rtol * b_norm) / beta, cs=cs)
            try:
                req = next(inner)
                while True:
                    if req[0] == "mv":
                        mv(req[1])
                    req = inner.send((yield req))
            except StopIteration as stop:
                Q, R, B, vs, y, pres = stop.value
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
    return x, j_outer + 1


def gcrotmk_device(ctx, matvec, b, n, rtol=1e-5, atol=0.0, maxiter=1000, m=20, k=None, complex_pairs=False, x0=None,
                   cols_per_pass=1, ops=None):
    """Solve A x = b; ``matvec(buf) -> new buf`` applies A on the device.

    With ``complex_pairs`` the vectors are (re, im) pairs of device buffers and the arithmetic is
    complex (SciPy's gcrotmk on a complex LinearOperator).  ``cols_per_pass``: the Arnoldi sweep's columns per pass
    (1 = SciPy's order of rounding, 4 = blocked; hipeig.h).  Returns ``(x, info, stats)`` with SciPy's ``info`` convention."""
    if ops is None:                                       # ``ops``: another vector-operation provider (the CPU tests pass a NumPy one)
        ops = _PairOps(ctx, n, cols_per_pass) if complex_pairs else _Ops(ctx, n, cols_per_pass)
    stats = {"outer": 0, "matvecs": 0}
    gen = _gcrotmk(ops, ctx, b, n, rtol, atol, maxiter, m, k, complex_pairs, x0, stats)
    try:
        req = next(gen)
        while True:
            req = gen.send(matvec(req[1]) if req[0] == "mv" else ops.arnoldi_step(req[1], req[2]))
    except StopIteration as stop:
        x, info = stop.value
    return x, info, stats


def gcrotmk_device_block(ctx, block_matvec, bs, n, rtol=1e-5, atol=0.0, maxiter=1000, m=20, k=None, complex_pairs=False,
                         cols_per_pass=1, ops_factory=None):
    """The solves ``A x_i = b_i`` for several right-hand sides advanced in LOCK STEP: every solve is the solver above,
    unchanged - its own Krylov spaces, its own recycle pairs, its own stopping - but their operator applications are
    collected and handed to ``block_matvec([v_0, v_1, ...]) -> [A v_0, A v_1, ...]`` together, so that they run as block
    products (one pass over the operator for several vectors).  The right-hand sides of one FEAST contour point share
    operator and shift (feast.py:198-200).  A solve that has finished simply drops out of the block.
    Returns ``[(x, info, stats), ...]`` in the order of ``bs``."""
    results = [None] * len(bs)
    stats = [{"outer": 0, "matvecs": 0} for _ in bs]
    gens, opss, req = [], [], {}

    def advance(i, value=None, first=False):
        try:
            req[i] = next(gens[i]) if first else gens[i].send(value)
        except StopIteration as stop:
            results[i] = stop.value + (stats[i],)
            req.pop(i, None)

    for i, b in enumerate(bs):
        if ops_factory is not None:
            ops = ops_factory()
        else:
            ops = _PairOps(ctx, n, cols_per_pass) if complex_pairs else _Ops(ctx, n, cols_per_pass)
        opss.append(ops)
        gens.append(_gcrotmk(ops, ctx, b, n, rtol, atol, maxiter, m, k, complex_pairs, None, stats[i]))
        advance(i, first=True)
    split = complex_pairs and all(hasattr(o, "arnoldi_begin") for o in opss)      # the split (begin / end) step exists for device pairs
    batched = split and ops_factory is None and n <= _PairOps.BATCH_MAX_N and all(o.cols_per_pass in (1, 4) for o in opss)
    while req:
        # the orthogonalisation steps of all right-hand sides that wait for one: enqueued back to back, collected
        # afterwards, so that the host work of one (QR insert, bookkeeping) runs under the kernels of the next
        arn = sorted(i for i, r in req.items() if r[0] == "arn")
        while arn:
            batch = arn[:16]
            if split and all(len(req[i][1]) <= _PairOps.SPLIT_MAX_COLS for i in batch):
                if not (batched and len(batch) > 1 and
                        _PairOps.arnoldi_begin_batch([opss[i] for i in batch], [(req[i][1], req[i][2]) for i in batch])):
                    for slot, i in enumerate(batch):
                        opss[i].arnoldi_begin(req[i][1], req[i][2], slot)
                for slot, i in enumerate(batch):
                    advance(i, opss[i].arnoldi_end(len(req[i][1]), slot))
            else:
                for i in batch:
                    advance(i, opss[i].arnoldi_step(req[i][1], req[i][2]))
            arn = sorted(i for i, r in req.items() if r[0] == "arn")
        idx = sorted(req)                                     # everybody left waits for a product
        if not idx:
            break
        outs = block_matvec([req[i][1] for i in idx])
        for i, o in zip(idx, outs):
            advance(i, o)
    return results

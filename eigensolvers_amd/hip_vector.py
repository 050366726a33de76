"""``HipVector``: the MI355X backend of the ``AbstractVector`` plugin surface.

It sits beside the reference's ``NumpyVector`` (numpyVector.py:23-238): same constructor
shape ``HipVector(array, options)``, same ``options["linearSystemArgs"]`` keys, same
methods and static hooks, same error behaviour - but the data is a device-resident fp64
buffer and every operation is a call into ``libhipeig.so`` (hand-written gfx950 kernels).
The operator handed to the solver must be a ``HipCsrOperator`` (device-resident CSR);
``HipCsrOperator.from_scipy`` / ``from_dense`` / ``generate`` create one.

With a communicator attached to the context (``HipContext.attach_comm``) a ``HipVector``
holds the LOCAL row slice of a row-partitioned global vector and the same calls become
collective: reductions all-reduce, operator applications all-gather the operand.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from .abstract_vector import AbstractVector, LINDEP_DEFAULT_VALUE

_ORTHO_METHODS = {"mgs": 0, "cgs2": 1}


def _ptr_table(bufs):
    arr = (C.c_void_p * len(bufs))(*[b.ptr for b in bufs])
    return C.cast(arr, C.POINTER(C.c_void_p)), arr


class HipContext:
    """Owns the device context handle (streams, workspaces, optional RCCL communicator)."""

    _default = None

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        h = C.c_void_p()
        _lib.call("hipeig_ctx_create", int(device), C.byref(h))
        self.handle = h
        self.device = int(device)
        self.nranks, self.rank = 1, 0
        self.direct_only = False         # a communicator without RCCL (attach_direct_only): no exchange for block operands
        self.partitioned = True          # False in replica mode (set_partitioned): whole operators / vectors on every rank
        self._pool = {}
        self._finalizer = weakref.finalize(self, HipContext._destroy, h, self._pool)

    @staticmethod
    def _destroy(handle, pool):
        lib = _lib.load()
        for ptrs in pool.values():
            for p in ptrs:
                lib.hipeig_vec_free(handle, C.c_void_p(p))
        pool.clear()
        lib.hipeig_ctx_destroy(handle)

    @classmethod
    def default(cls):
        if cls._default is None:
            cls._default = cls()
        return cls._default

    # ---- communicator -------------------------------------------------------------
    @staticmethod
    def new_unique_id():
        buf = C.create_string_buffer(128)
        _lib.call("hipeig_comm_unique_id", C.cast(buf, C.c_void_p))
        return bytes(buf.raw)

    def attach_comm(self, nranks, rank, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _lib.call("hipeig_comm_init", self.handle, int(nranks), int(rank), C.cast(buf, C.c_void_p))
        self.nranks, self.rank = int(nranks), int(rank)

    def attach_loopback(self, group_handle, nranks, rank):
        """Join an in-process loopback group (``distributed.LoopbackGroup``)."""
        _lib.call("hipeig_comm_init_loopback", self.handle, group_handle, int(rank))
        self.nranks, self.rank = int(nranks), int(rank)

    # ---- operand exchange of a row-partitioned product -------------------------------------------
    GATHER_BACKENDS = {0: "rccl", 1: "direct"}

    def direct_alloc(self, capacity_doubles):
        """This rank's buffers for the direct exchange; returns its 192-byte record (two hipIpc handles + PCI bus id)."""
        buf = C.create_string_buffer(192)
        _lib.call("hipeig_direct_alloc", self.handle, int(capacity_doubles), C.cast(buf, C.c_void_p))
        return bytes(buf.raw)

    def direct_attach(self, records):
        """Map the peers' buffers: ``records`` = every rank's 192-byte record in rank order."""
        blob = b"".join(records)
        buf = C.create_string_buffer(blob, len(blob))
        _lib.call("hipeig_direct_attach", self.handle, C.cast(buf, C.c_void_p))

    def direct_release(self):
        """Free the buffers of the direct exchange (after a collective fall-back to RCCL)."""
        _lib.call("hipeig_direct_release", self.handle)

    def set_direct_wait_limit(self, seconds):
        """Wall-clock limit of the direct exchange's bounded waits from now on (<= 0: the default)."""
        _lib.call("hipeig_comm_set_wait_limit", self.handle, float(seconds))

    def set_exchange(self, on):
        """Measurement aid: ``False`` makes this context's products skip the operand exchange (own slice only; results
        meaningless) so that one rank's sweeps can be timed alone; switch back before the next collective product."""
        _lib.call("hipeig_comm_set_exchange", self.handle, 1 if on else 0)

    def attach_direct_only(self, nranks, rank):
        """A communicator without RCCL (rank / size only): every exchange goes through the direct peer-write backend,
        which must be attached before the first operator is created (``distributed.attach_direct``)."""
        _lib.call("hipeig_comm_init_direct", self.handle, int(nranks), int(rank))
        self.nranks, self.rank = int(nranks), int(rank)
        self.direct_only = True

    def set_gather_backend(self, name):
        code = {v: k for k, v in self.GATHER_BACKENDS.items()}[name]
        _lib.call("hipeig_comm_set_gather_backend", self.handle, code)

    def set_allreduce_backend(self, name):
        code = {v: k for k, v in self.GATHER_BACKENDS.items()}[name]
        _lib.call("hipeig_comm_set_allreduce_backend", self.handle, code)

    def set_gather_chunks(self, nchunks):
        """Chunks of the operand exchange (1-4; 0 = automatic) for operators created from now on (same on every rank)."""
        _lib.call("hipeig_comm_set_gather_chunks", self.handle, int(nchunks))

    def gather_info(self):
        info = (C.c_int64 * 8)()
        _lib.call("hipeig_comm_gather_info", self.handle, info)
        return {"backend": self.GATHER_BACKENDS[int(info[0])], "direct_attached": bool(info[1]), "capacity": int(info[2]),
                "exchanges": int(info[3]), "wait_error": int(info[4]), "chunks_override": int(info[5]),
                "allreduce_backend": self.GATHER_BACKENDS[int(info[6])]}

    def phase_timing(self, on):
        _lib.call("hipeig_phase_timing", self.handle, 1 if on else 0)

    def phase_times(self):
        """ms of the most recent partitioned product: exchange, own-window sweep, remaining sweep, whole product, idle gap
        (None where the phase did not occur)."""
        out = (C.c_double * 8)()
        _lib.call("hipeig_phase_get", self.handle, out)
        names = ("gather_ms", "local_sweep_ms", "remote_sweep_ms", "product_ms", "idle_before_first_chunk_ms")
        return {k: (float(out[i]) if out[i] >= 0 else None) for i, k in enumerate(names)}

    def allreduce_ms(self, count=2, reps=50):
        out = C.c_double()
        _lib.call("hipeig_comm_bench_allreduce", self.handle, int(count), int(reps), C.byref(out))
        return float(out.value)

    def set_partitioned(self, flag):
        """False: replica mode - whole operators and vectors on every rank, no implicit collectives;
        only ``allreduce_vector`` exchanges data (FEAST contour replicas)."""
        _lib.call("hipeig_comm_set_partitioned", self.handle, 1 if flag else 0)
        self.partitioned = bool(flag)

    def allreduce_vector(self, buf):
        """SUM of a device buffer over the ranks, in place."""
        _lib.call("hipeig_vec_allreduce", self.handle, buf.ptr, buf.n)

    # ---- memory -------------------------------------------------------------------
    def alloc(self, n):
        free = self._pool.get(n)
        if free:
            return DeviceBuffer(self, free.pop(), n)
        p = C.c_void_p()
        _lib.call("hipeig_vec_alloc", self.handle, int(n), C.byref(p))
        return DeviceBuffer(self, p.value, n)

    def synchronize(self):
        _lib.call("hipeig_ctx_sync", self.handle)

    def device_info(self):
        info = (C.c_int64 * 8)()
        name = C.create_string_buffer(256)
        _lib.call("hipeig_device_info", self.handle, info, name, 256)
        return {"name": name.value.decode(), "cus": info[0], "wave": info[1], "hbm_total": info[2],
                "hbm_free": info[3], "l2_bytes": info[4]}

    def timer_start(self):
        _lib.call("hipeig_timer_start", self.handle)

    def timer_stop(self):
        ms = C.c_float()
        _lib.call("hipeig_timer_stop", self.handle, C.byref(ms))
        return float(ms.value)


class DeviceBuffer:
    """A device allocation of n doubles; returned to the context's pool when collected.

    All work is ordered on one stream, so recycling a buffer without a sync is safe."""

    __slots__ = ("ctx", "ptr", "n", "__weakref__")

    def __init__(self, ctx, ptr, n):
        self.ctx, self.ptr, self.n = ctx, ptr, n

    def __del__(self):
        try:
            self.ctx._pool.setdefault(self.n, []).append(self.ptr)
        except Exception:
            pass


class HipCsrOperator:
    """Device-resident sparse symmetric operator (local rows of a row partition)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.handle = handle
        info = (C.c_int64 * 8)()
        _lib.call("hipeig_csr_info", handle, info)
        self.nrows, self.ncols, self.nnz, self.row_offset = info[0], info[1], info[2], info[3]
        self.device_bytes = info[5]
        self.shape = (self.nrows, self.ncols)
        self.dtype = np.dtype(np.float64)
        self._finalizer = weakref.finalize(self, HipCsrOperator._destroy, ctx, handle)

    @staticmethod
    def _destroy(ctx, handle):
        _lib.load().hipeig_csr_destroy(ctx.handle, handle)

    @classmethod
    def from_csr_arrays(cls, rowptr, col, val, ncols, row_offset=0, ctx=None):
        ctx = ctx or HipContext.default()
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        h = C.c_void_p()
        _lib.call("hipeig_csr_create", ctx.handle, len(rowptr) - 1, int(ncols), int(row_offset),
                  rowptr.ctypes.data_as(C.POINTER(C.c_int64)), col.ctypes.data_as(C.POINTER(C.c_int32)),
                  val.ctypes.data_as(C.POINTER(C.c_double)), C.byref(h))
        return cls(ctx, h)

    @classmethod
    def from_scipy(cls, A, row_begin=0, row_end=None, ctx=None):
        """Upload rows [row_begin,row_end) of a scipy.sparse matrix (real, any format)."""
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        if np.iscomplexobj(A.data):
            raise TypeError("HipCsrOperator holds real fp64 operators")
        row_end = A.shape[0] if row_end is None else row_end
        sl = A[row_begin:row_end]
        return cls.from_csr_arrays(sl.indptr, sl.indices, sl.data, A.shape[1], row_begin, ctx)

    @classmethod
    def from_dense(cls, M, ctx=None):
        """Dense ndarray stored as a full CSR (every entry kept) - plumbing-size problems."""
        M = np.asarray(M, dtype=np.float64)
        n, m = M.shape
        rowptr = np.arange(0, n * m + 1, m, dtype=np.int64)
        col = np.tile(np.arange(m, dtype=np.int32), n)
        return cls.from_csr_arrays(rowptr, col, M.ravel(), m, 0, ctx)

    @classmethod
    def generate(cls, N, nnz_row=64, seed=7, row_begin=0, row_end=None, ctx=None, **kw):
        """Synthetic gapped random-sparse Hermitian operator built on the device
        (bit-identical to ``generators.gapped_csr_host``)."""
        from .generators import gapped_params
        ctx = ctx or HipContext.default()
        p = gapped_params(N, nnz_row, seed, **kw)
        row_end = N if row_end is None else row_end
        t = np.ascontiguousarray(p["targets"], dtype=np.float64)
        h = C.c_void_p()
        _lib.call("hipeig_csr_generate", ctx.handle, int(N), int(row_begin), int(row_end), p["K"],
                  C.c_uint64(p["seed"]), p["eps"], C.c_uint32(p["thresh24"]),
                  t.ctypes.data_as(C.POINTER(C.c_double)), len(t), C.byref(h))
        return cls(ctx, h)

    VARIANTS = {0: "none", 1: "csr-vector", 2: "csr-stream", 3: "column-window-blocked(wave)",
                4: "column-window-blocked(workgroup)", 5: "column-window-blocked(workgroup, fixed-point)"}

    def set_variant(self, variant):
        """Operator kernel: 0 automatic; 1 CSR-vector, 2 CSR-stream, 3 column-window blocked with wave-owned rows
        (these three fix the order of the adds inside a row: bitwise reproducible); 4 column-window blocked with
        workgroup-owned rows and fp64 LDS atomics (the fast default for large operators; a row's sum is
        reproducible to rounding only); 5 the same sweep with FIXED-POINT accumulators - integer adds commute, so
        it is bitwise reproducible at the speed of 4, with an absolute error per row of
        ~nnz_row * 2^-61 * max_i sum_j|a_ij| * max|x| instead of fp64's relative one."""
        _lib.call("hipeig_csr_set_variant", self.handle, int(variant))
        self._variant = int(variant)

    def honour_reduction_option(self, options):
        """``options["reduction"]`` of the vectors: "deterministic" restricts the operator to
        bitwise reproducible kernels - on the automatic choice CSR-stream stays, the fixed-point variant 5 takes the
        place of the fp64-atomic variant 4 (same speed) and block products take the row-owner kernel; an operator
        pinned to 4 moves to 5.  "fast" undoes both.  An explicitly pinned variant 1-3 is left alone."""
        want = (options or {}).get("reduction")
        if want not in (None, "deterministic", "fast"):
            raise ValueError(f'options["reduction"] must be "deterministic" or "fast", got {want!r}')
        if want is None or want == getattr(self, "_reduction", None):
            return
        if getattr(self, "_reduction_pinned", False):
            return                                   # set_reduction() on the operator wins over per-vector options
        if getattr(self, "_reduction", None) is not None:
            # the mode is a property of the OPERATOR (its kernels), shared by every vector that uses it: two vectors asking
            # for different modes would flip the kernels back and forth under each other
            raise ValueError(f'this operator already runs with reduction={self._reduction!r} (asked for by another vector); '
                             f'a vector with reduction={want!r} needs its own operator, or settle it with H.set_reduction()')
        cur = getattr(self, "_variant", 0)
        _lib.call("hipeig_csr_set_reproducible", self.handle, 1 if want == "deterministic" else 0)
        if want == "deterministic" and cur == 4:
            self.set_variant(5)
            self._moved_by_option = True
        elif want == "fast" and cur == 5 and getattr(self, "_moved_by_option", False):
            self.set_variant(4)
            self._moved_by_option = False
        self._reduction = want

    def set_reduction(self, mode):
        """Pin the reduction mode of this operator ("deterministic": bitwise reproducible kernels only; "fast"; None: back to
        what the vectors' options ask for) whatever ``options["reduction"]`` later vectors carry."""
        self._reduction_pinned = False
        previous, self._reduction = getattr(self, "_reduction", None), None
        if mode is None:
            if previous == "deterministic":              # back to the unrestricted kernels, then forget
                self.honour_reduction_option({"reduction": "fast"})
                self._reduction = None
            return
        self.honour_reduction_option({"reduction": mode})
        self._reduction_pinned = True

    def fixed_point_info(self):
        """(max_i sum_j |a_ij|, max |x| of the last variant-5 operand): what bounds variant 5's absolute error."""
        out = (C.c_double * 2)()
        _lib.call("hipeig_csr_fixed_info", self.ctx.handle, self.handle, out)
        return float(out[0]), float(out[1])

    def last_variant(self):
        """Name of the kernel variant the most recent product used."""
        info = (C.c_int64 * 8)()
        _lib.call("hipeig_csr_info", self.handle, info)
        return self.VARIANTS[int(info[4])]

    BLOCK_VARIANTS = {0: "none", 1: "row-owner", 2: "column-window-blocked"}

    def set_block_variant(self, variant):
        """Kernel of the block product / block solve: 0 automatic, 1 row-owner CSR, 2 window-blocked."""
        _lib.call("hipeig_csr_set_block_variant", self.handle, int(variant))

    def block_info(self):
        info = (C.c_int64 * 4)()
        _lib.call("hipeig_csr_block_info", self.handle, info)
        return {"variant": self.BLOCK_VARIANTS[int(info[0])], "row_blocks": int(info[1]), "windows": int(info[2]),
                "rows_per_block": int(info[3])}

    def layout_info(self):
        """Layout constants of the blocked copy the last product ran on (a counter profile is valid for these only)."""
        out = (C.c_int64 * 12)()
        _lib.call("hipeig_csr_layout_info", self.handle, out)
        keys = ("variant", "rows_per_block", "window_bits", "row_blocks", "windows", "column_splits", "workgroups_per_launch",
                "threads", "unroll", "exchange_chunks", "rows_per_rank_chunk", "bins")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def launches_per_apply(self):
        """Kernel launches (sweeps) one product takes with the variant last used."""
        info = (C.c_int64 * 8)()
        _lib.call("hipeig_csr_info", self.handle, info)
        return max(1, int(info[7]))

    def to_scipy(self):
        import scipy.sparse as sp
        rp = np.empty(self.nrows + 1, dtype=np.int64)
        col = np.empty(self.nnz, dtype=np.int32)
        val = np.empty(self.nnz, dtype=np.float64)
        _lib.call("hipeig_csr_download", self.ctx.handle, self.handle,
                  rp.ctypes.data_as(C.POINTER(C.c_int64)), col.ctypes.data_as(C.POINTER(C.c_int32)),
                  val.ctypes.data_as(C.POINTER(C.c_double)))
        return sp.csr_matrix((val, col, rp), shape=(self.nrows, self.ncols))

    def algorithmic_bytes(self):
        """nnz*(8+4) + (nrows+1)*4 + 8*ncols (x once) + 8*nrows (y)."""
        return self.nnz * 12 + (self.nrows + 1) * 4 + 8 * self.ncols + 8 * self.nrows

    # y = H x on raw buffers
    def apply(self, x, y):
        _lib.call("hipeig_spmv", self.ctx.handle, self.handle, x.ptr, y.ptr)

    def apply_block(self, xs):
        """[H x for x in xs] as one block product (tall-skinny SpMM); xs are DeviceBuffers."""
        ys = [self.ctx.alloc(self.nrows) for _ in xs]
        xt, keep1 = _ptr_table(xs)
        yt, keep2 = _ptr_table(ys)
        _lib.call("hipeig_spmm", self.ctx.handle, self.handle, len(xs), xt, yt)
        return ys

    def apply_shifted(self, sigma, x, y, reverse=False):
        _lib.call("hipeig_spmv_shift", self.ctx.handle, self.handle, float(sigma),
                  -1.0 if reverse else 1.0, x.ptr, y.ptr)

    def apply_shifted_pair(self, z, xr, xi, yr, yi, reverse=False):
        """(yr, yi) = sign*(z*x - H x) for the complex operand x = xr + i xi and the complex shift z (the operator is
        real): one sweep of the operator for both halves on large operators (``hipeig_spmv_shift_pair``)."""
        z = complex(z)
        _lib.call("hipeig_spmv_shift_pair", self.ctx.handle, self.handle, z.real, z.imag,
                  -1.0 if reverse else 1.0, xr.ptr, xi.ptr, yr.ptr, yi.ptr)

    def apply_shifted_pairs(self, z, xs, reverse=False):
        """[sign*(z*x - H x) for x in xs] for complex operands xs = [(re, im) DeviceBuffers, ...]: four operands share one
        pass over the operator (``hipeig_spmm_shift_pairs``).  Returns new (re, im) buffer pairs."""
        z = complex(z)
        n = self.nrows
        ys = [(self.ctx.alloc(n), self.ctx.alloc(n)) for _ in xs]
        t_xr, k1 = _ptr_table([x[0] for x in xs])
        t_xi, k2 = _ptr_table([x[1] for x in xs])
        t_yr, k3 = _ptr_table([y[0] for y in ys])
        t_yi, k4 = _ptr_table([y[1] for y in ys])
        _lib.call("hipeig_spmm_shift_pairs", self.ctx.handle, self.handle, len(xs), z.real, z.imag, -1.0 if reverse else 1.0,
                  t_xr, t_xi, t_yr, t_yi)
        return ys

    def apply_pair(self, xr, xi, yr, yi):
        """(yr, yi) = (H xr, H xi): ``applyOp`` on a complex vector, one sweep where the pair kernel applies."""
        _lib.call("hipeig_spmv_shift_pair", self.ctx.handle, self.handle, 0.0, 0.0, 0.0, xr.ptr, xi.ptr, yr.ptr, yi.ptr)

    def pair_info(self):
        """{'fused': the most recent pair product ran as one sweep, 'launches': its sweep launches}"""
        out = (C.c_int64 * 2)()
        _lib.call("hipeig_csr_pair_info", self.handle, out)
        return {"fused": bool(out[0]), "launches": int(out[1])}

    def __matmul__(self, v):
        if isinstance(v, HipVector):
            return v.applyOp(self)
        return NotImplemented


class HipVector(AbstractVector):
    """Device-resident fp64 vector with the NumpyVector interface.  A complex array gives a
    ``HipComplexVector`` (two real halves), the counterpart of a complex-dtype NumpyVector."""

    def __new__(cls, array=None, options=None, ctx=None):
        if cls is HipVector and not isinstance(array, DeviceBuffer) and np.iscomplexobj(array):
            return HipComplexVector(array, options, ctx=ctx)
        return super().__new__(cls)

    def __init__(self, array, options=None, ctx=None):
        given = {} if options is None else options
        if isinstance(array, DeviceBuffer):
            self._buf = array
            self.ctx = array.ctx
        else:
            host = np.ascontiguousarray(array, dtype=np.float64)
            if host.ndim != 1:
                raise ValueError("HipVector holds 1-D vectors")
            if np.iscomplexobj(array):               # only reachable from a subclass: HipVector(...) itself dispatches
                raise TypeError("a real HipVector cannot hold complex data; use HipComplexVector")
            self.ctx = ctx or HipContext.default()
            self._buf = self.ctx.alloc(host.size)
            _lib.call("hipeig_vec_upload", self.ctx.handle, self._buf.ptr,
                      host.ctypes.data_as(C.c_void_p), host.size)
        self.size = self._buf.n
        self.shape = (self._buf.n,)
        # numpyVector.py:29-36: defaults are written back into the caller's dict, which all
        # vectors derived from this one then share
        lsa = given.get("linearSystemArgs", dict())
        lsa.setdefault("linearSolver", "minres")
        lsa.setdefault("linearIter", 1000)
        lsa.setdefault("linear_tol", 1e-4)
        lsa.setdefault("linear_atol", 1e-4)
        self.options = {"linearSystemArgs": lsa}
        for extra in ("orthogonalization", "blockSolve", "reduction"):
            if extra in given:
                self.options[extra] = given[extra]
        self.last_solve_stats = None

    # ---- helpers -------------------------------------------------------------------
    def _new(self, buf):
        return HipVector(buf, self.options)

    @staticmethod
    def fromArray(template, array):
        """A vector like ``template`` (same context, shared options) from host data - the hook
        ``checkpoint.restore_vectors`` uses."""
        return HipVector(array, template.options, ctx=template.ctx)

    @property
    def array(self):
        """Host copy of the (local) data - the reference's tests read ``.array``."""
        out = np.empty(self._buf.n, dtype=np.float64)
        _lib.call("hipeig_vec_download", self.ctx.handle, out.ctypes.data_as(C.c_void_p),
                  self._buf.ptr, self._buf.n)
        return out

    # ---- properties ----------------------------------------------------------------
    @property
    def hasExactAddition(self):
        return True

    @property
    def dtype(self):
        return np.dtype(np.float64)

    @property
    def maxD(self):
        return 0

    # ---- arithmetic (out of place, numpyVector.py:57-64) ----------------------------
    def _scaled(self, alpha):
        if isinstance(alpha, complex) or np.iscomplexobj(alpha):
            alpha = complex(alpha)                      # feast.py:91-92: mult * Qe with complex mult
            return HipComplexVector(self._scaled(alpha.real), self._scaled(alpha.imag))
        out = self.ctx.alloc(self._buf.n)
        _lib.call("hipeig_scale", self.ctx.handle, self._buf.n, float(alpha), self._buf.ptr, out.ptr)
        return self._new(out)

    def __mul__(self, other):
        return self._scaled(other)

    def __rmul__(self, other):
        return self._scaled(other)

    def __truediv__(self, other):
        out = self.ctx.alloc(self._buf.n)
        _lib.call("hipeig_divide", self.ctx.handle, self._buf.n, float(other), self._buf.ptr, out.ptr)
        return self._new(out)

    def __imul__(self, other):
        raise NotImplementedError          # numpyVector.py:66-67

    def __itruediv__(self, other):
        raise NotImplementedError          # numpyVector.py:69-70

    def __len__(self):
        return self._buf.n

    # ---- instance methods -----------------------------------------------------------
    def normalize(self):
        nrm = C.c_double()
        _lib.call("hipeig_normalize", self.ctx.handle, self._buf.n, self._buf.ptr, C.byref(nrm))
        return self

    def norm(self):
        out = C.c_double()
        _lib.call("hipeig_nrm2", self.ctx.handle, self._buf.n, self._buf.ptr, C.byref(out))
        return float(out.value)

    def real(self):
        # the FEAST driver calls ``typeClass.real(mult * Qe)`` unbound, with a complex operand
        if isinstance(self, HipComplexVector):
            return self.re.copy()
        return self.copy()

    def conjugate(self):
        if isinstance(self, HipComplexVector):
            return HipComplexVector(self.re.copy(), self.im * -1.0)
        return self.copy()

    def vdot(self, other, conjugate=True):
        out = C.c_double()
        _lib.call("hipeig_dot", self.ctx.handle, self._buf.n, self._buf.ptr, other._buf.ptr, C.byref(out))
        return float(out.value)

    def copy(self):
        out = self.ctx.alloc(self._buf.n)
        _lib.call("hipeig_vec_copy", self.ctx.handle, out.ptr, self._buf.ptr, self._buf.n)
        return self._new(out)

    def applyOp(self, other):
        if not isinstance(other, HipCsrOperator):
            raise TypeError("HipVector.applyOp needs a HipCsrOperator (device-resident CSR)")
        other.honour_reduction_option(self.options)
        out = self.ctx.alloc(other.nrows)
        other.apply(self._buf, out)
        return self._new(out)

    def compress(self):
        return self

    # ---- static hooks ----------------------------------------------------------------
    @staticmethod
    def linearCombination(vectors, coeffs):
        assert len(vectors) == len(coeffs)
        v0 = vectors[0]
        cf = np.ascontiguousarray(coeffs, dtype=np.float64)
        out = v0.ctx.alloc(v0._buf.n)
        tab, keep = _ptr_table([v._buf for v in vectors])
        _lib.call("hipeig_lincomb", v0.ctx.handle, v0._buf.n, len(vectors),
                  cf.ctypes.data_as(C.POINTER(C.c_double)), tab, out.ptr)
        return v0._new(out)

    @staticmethod
    def linearCombinationBlock(vectors, coeffs):
        """k combinations at once: out[c] = sum_j coeffs[j, c] * vectors[j]
        (basisTransformation with a coefficient matrix, util_funcs.py:229-230)."""
        v0 = vectors[0]
        cm = np.ascontiguousarray(coeffs, dtype=np.float64)
        m, k = cm.shape
        assert m == len(vectors)
        outs = [v0.ctx.alloc(v0._buf.n) for _ in range(k)]
        tab, keep = _ptr_table([v._buf for v in vectors])
        otab, okeep = _ptr_table(outs)
        _lib.call("hipeig_lincomb_block", v0.ctx.handle, v0._buf.n, m, k,
                  cm.ctypes.data_as(C.POINTER(C.c_double)), k, tab, otab)
        return [v0._new(o) for o in outs]

    @staticmethod
    def orthogonalize_against_set(x, qs, lindep=LINDEP_DEFAULT_VALUE):
        """numpyVector.py:121-145.  Default: the reference's own sweep - sequential modified Gram-Schmidt
        with the division by q.q, ``None`` when what is left has x.x <= lindep - so the linear-dependency
        exit fires where the reference's does.  ``options["orthogonalization"] = "cgs2"`` selects two
        batched classical passes instead (one reduction per pass; its second pass changes the inner
        product compared with ``lindep``, so that exit may fire at another iteration)."""
        new = x.copy()
        method = _ORTHO_METHODS[x.options.get("orthogonalization", "mgs")]
        ip = C.c_double()
        dep = C.c_int()
        tab, keep = _ptr_table([q._buf for q in qs]) if len(qs) else (None, None)
        _lib.call("hipeig_orthonormalize", x.ctx.handle, new._buf.n, len(qs), tab, new._buf.ptr,
                  float(lindep), method, C.byref(ip), C.byref(dep))
        return None if dep.value else new

    @staticmethod
    def solve(H, b, sigma, x0=None, opType="her", reverseGF=False):
        if not isinstance(H, HipCsrOperator):
            raise TypeError("HipVector.solve needs a HipCsrOperator (device-resident CSR)")
        if isinstance(b, HipComplexVector):                 # HipVector(complex array) is a HipComplexVector: same entry point
            return HipComplexVector.solve(H, b, sigma, x0, opType, reverseGF)
        if x0 is not None and not isinstance(x0, (HipVector, HipComplexVector)):
            x0 = HipVector(np.asarray(x0), ctx=b.ctx)      # NumpyVector hands an ndarray on to SciPy; complex arrays become HipComplexVector
        if isinstance(x0, HipComplexVector) and not (isinstance(sigma, complex) or np.iscomplexobj(sigma)):
            return HipComplexVector.solve(H, HipComplexVector(b, b * 0.0), sigma, x0, opType, reverseGF)    # a complex guess makes the solve complex
        H.honour_reduction_option(b.options)
        o = b.options["linearSystemArgs"]
        name = o["linearSolver"]
        if name == "pardiso":
            return HipVector._solve_exact_small(H, b, sigma, reverseGF)
        if isinstance(sigma, complex) or np.iscomplexobj(sigma):
            return HipVector._solve_complex(H, b, complex(sigma), o, reverseGF, x0)
        if name == "gcrotmk":
            # numpyVector.py:161: gcrotmk(linOp, b, x0, tol, atol, maxiter) with SciPy's m = k = 20
            from .gcrotmk import gcrotmk_device

            def matvec(buf):
                out = b.ctx.alloc(b._buf.n)
                H.apply_shifted(sigma, buf, out, reverse=reverseGF)
                return out

            xbuf, conv, gstats = gcrotmk_device(b.ctx, matvec, b._buf, b._buf.n, rtol=float(o["linear_tol"]),
                                                atol=float(o["linear_atol"]), maxiter=int(o["linearIter"]),
                                                x0=None if x0 is None else x0._buf,
                                                cols_per_pass=int(o.get("arnoldiColumnsPerPass", 1)))
            res = b._new(xbuf)
            res.last_solve_stats = b.last_solve_stats = {"iterations": gstats["matvecs"], "outer": gstats["outer"]}
            if conv != 0:
                raise UserWarning("Warning:: Iterative solver is not converged ")
            return res
        if name != "minres":
            raise Exception("Got linear solver other than gcrotmk, minres and pardiso!")
        out = b.ctx.alloc(b._buf.n)
        info = C.c_int()
        stats = (C.c_double * 8)()
        if x0 is None:
            _lib.call("hipeig_minres", b.ctx.handle, H.handle, float(sigma), -1.0 if reverseGF else 1.0,
                      b._buf.ptr, out.ptr, float(o["linear_tol"]), int(o["linearIter"]), C.byref(info), stats)
        else:               # scipy.sparse.linalg.minres(linOp, b, x0): r1 = b - A x0, the iterate starts at x0
            _lib.call("hipeig_minres_x0", b.ctx.handle, H.handle, float(sigma), -1.0 if reverseGF else 1.0,
                      b._buf.ptr, x0._buf.ptr, out.ptr, float(o["linear_tol"]), int(o["linearIter"]), C.byref(info), stats)
        res = b._new(out)
        res.last_solve_stats = {"iterations": int(stats[0]), "istop": int(stats[1]), "rnorm": stats[2],
                                "Anorm": stats[3], "ynorm": stats[4], "test1": stats[5],
                                "test2": stats[6], "Acond": stats[7]}
        if b.ctx.nranks > 1 or os.environ.get("HIPEIG_FORCE_COLLECTIVES", "0") not in ("", "0"):
            cs = (C.c_int64 * 4)()
            _lib.call("hipeig_comm_stats", b.ctx.handle, cs)
            res.last_solve_stats["collectives"] = int(cs[0])
        b.last_solve_stats = res.last_solve_stats
        if info.value != 0:
            # numpyVector.py:175-177: the warning is escalated to an exception
            raise UserWarning("Warning:: Iterative solver is not converged ")
        return res

    BLOCK_SOLVE_MIN = 3      # fewer right-hand sides are solved one by one (measured at N = 1e6: 2 columns 0.97x, 3: 1.6x, 4: 1.95x, 8: 2.8x)

    @staticmethod
    def solveBlock(H, bs, sigma, x0=None, opType="her", reverseGF=False):
        """``[solve(H, b, sigma) for b in bs]`` with the solves advanced in lock step: one block product
        per MINRES iteration for up to 8 right-hand sides (inexact_Lanczos.py:319-320 calls ``solve``
        once per block vector on the same operator and shift).  Every column runs the recurrences and
        stopping tests of the single solve; results, ``last_solve_stats`` and the exception on
        non-convergence (numpyVector.py:175-177) are those of the one-by-one calls.  Blocks of <= 4 use a
        4-wide interleave (twice the rows per workgroup), larger ones chunks of 8.  Solvers other than MINRES,
        complex shifts and fewer than ``BLOCK_SOLVE_MIN`` right-hand sides take the one-by-one calls."""
        bs = list(bs)
        o = bs[0].options["linearSystemArgs"]
        if (o["linearSolver"] == "gcrotmk" and (isinstance(sigma, complex) or np.iscomplexobj(sigma)) and x0 is None
                and len(bs) >= 2 and isinstance(H, HipCsrOperator) and not bs[0].ctx.direct_only
                and (bs[0].ctx.nranks == 1 or not bs[0].ctx.partitioned)       # whole vectors: one GPU, or FEAST's contour replicas
                and all(isinstance(b, HipVector) for b in bs)):
            return HipVector._solve_complex_block(H, bs, complex(sigma), o, reverseGF)
        if (o["linearSolver"] != "minres" or isinstance(sigma, complex) or np.iscomplexobj(sigma)
                or x0 is not None or len(bs) < HipVector.BLOCK_SOLVE_MIN or not isinstance(H, HipCsrOperator)
                or bs[0].ctx.direct_only):
            return [HipVector.solve(H, b, sigma, x0, opType, reverseGF) for b in bs]
        H.honour_reduction_option(bs[0].options)
        ctx, n = bs[0].ctx, bs[0]._buf.n
        results = []
        for i0 in range(0, len(bs), 8):
            chunk = bs[i0:i0 + 8]
            k = len(chunk)
            outs = [ctx.alloc(n) for _ in chunk]
            bt, keep1 = _ptr_table([b._buf for b in chunk])
            xt, keep2 = _ptr_table(outs)
            info = (C.c_int * k)()
            stats = (C.c_double * (8 * k))()
            _lib.call("hipeig_minres_block", ctx.handle, H.handle, float(sigma), -1.0 if reverseGF else 1.0, k,
                      bt, xt, float(o["linear_tol"]), int(o["linearIter"]), info, stats)
            ncoll = None
            if ctx.nranks > 1 or os.environ.get("HIPEIG_FORCE_COLLECTIVES", "0") not in ("", "0"):
                cs = (C.c_int64 * 4)()
                _lib.call("hipeig_comm_stats", ctx.handle, cs)
                ncoll = int(cs[0])                 # collectives of the whole lock-step solve (all columns together)
            for j, b in enumerate(chunk):
                res = b._new(outs[j])
                st = stats[8 * j:8 * j + 8]
                res.last_solve_stats = b.last_solve_stats = {
                    "iterations": int(st[0]), "istop": int(st[1]), "rnorm": st[2], "Anorm": st[3],
                    "ynorm": st[4], "test1": st[5], "test2": st[6], "Acond": st[7]}
                if ncoll is not None:
                    res.last_solve_stats["collectives"] = ncoll
                results.append(res)
            if any(info[j] != 0 for j in range(k)):
                raise UserWarning("Warning:: Iterative solver is not converged ")
        return results

    EXACT_SOLVE_MAX = 96

    @staticmethod
    def _solve_exact_small(H, b, sigma, reverseGF):
        """``linearSolver="pardiso"`` (numpyVector.py:166-170: ``spsolve`` of ``sigma*I - H``, which the reference keeps
        "only for comparing with fortran"): Gaussian elimination with partial pivoting on the device, one workgroup,
        n <= 96.  Real or complex shift, real or complex right-hand side; never a CPU detour."""
        ctx, n = b.ctx, len(b)
        if n > HipVector.EXACT_SOLVE_MAX:
            raise NotImplementedError(f"linearSolver='pardiso' is the reference's small exact branch (Fortran comparison); "
                                      f"on the device it takes n <= {HipVector.EXACT_SOLVE_MAX}, got n = {n}")
        z = complex(sigma)
        cplx_b = isinstance(b, HipComplexVector)
        br = b.re if cplx_b else b
        bi = b.im._buf.ptr if cplx_b else None
        is_complex = cplx_b or z.imag != 0.0 or isinstance(sigma, complex) or np.iscomplexobj(sigma)
        xr = ctx.alloc(n)
        xi = ctx.alloc(n) if is_complex else None
        flag = C.c_int()
        _lib.call("hipeig_dense_solve_small", ctx.handle, H.handle, z.real, z.imag, -1.0 if reverseGF else 1.0,
                  br._buf.ptr, bi, xr.ptr, None if xi is None else xi.ptr, C.byref(flag))
        if flag.value:
            raise np.linalg.LinAlgError("sigma*I - H is singular to working precision")
        res = HipComplexVector(br._new(xr), br._new(xi)) if is_complex else br._new(xr)
        res.last_solve_stats = b.last_solve_stats = {"iterations": 0, "exact": True}
        return res

    @staticmethod
    def _solve_complex(H, b, z, o, reverseGF, x0=None):
        """(z*I - H) x = b with a complex contour point z, real H and real b (feast.py:83-90).
        The system is complex symmetric, so the reference uses GCROT there; the complex vectors
        are (re, im) pairs of device buffers and one complex product costs two operator sweeps."""
        if o["linearSolver"] != "gcrotmk":
            raise NotImplementedError("complex shifts need linearSolver='gcrotmk' on the device "
                                      f"(got {o['linearSolver']!r}; minres is for Hermitian systems)")
        from .gcrotmk import gcrotmk_device
        ctx, n = b.ctx, len(b)
        sgn = -1.0 if reverseGF else 1.0

        def matvec(v):                                  # sgn * ((zr + i zi)(vr + i vi) - H vr - i H vi)
            out_r, out_i = ctx.alloc(n), ctx.alloc(n)
            H.apply_shifted_pair(z, v[0], v[1], out_r, out_i, reverse=reverseGF)
            return (out_r, out_i)

        if isinstance(b, HipComplexVector):
            rhs = (b.re._buf, b.im._buf)
            b = b.re
        else:
            zero = ctx.alloc(n)
            _lib.call("hipeig_vec_fill", ctx.handle, zero.ptr, n, 0.0)
            rhs = (b._buf, zero)
        guess = None
        if x0 is not None:                              # SciPy's x0 (numpyVector.py:161): real or complex, array or vector
            if not isinstance(x0, (HipVector, HipComplexVector)):
                x0 = HipVector(np.asarray(x0), ctx=ctx)              # complex arrays come back as HipComplexVector
            if isinstance(x0, HipComplexVector):
                guess = (x0.re._buf, x0.im._buf)
            else:
                zero0 = ctx.alloc(n)
                _lib.call("hipeig_vec_fill", ctx.handle, zero0.ptr, n, 0.0)
                guess = (x0._buf, zero0)
        x, conv, gstats = gcrotmk_device(ctx, matvec, rhs, n, rtol=float(o["linear_tol"]),
                                         atol=float(o["linear_atol"]), maxiter=int(o["linearIter"]),
                                         complex_pairs=True, x0=guess,
                                         cols_per_pass=int(o.get("arnoldiColumnsPerPass", 1)))
        res = HipComplexVector(b._new(x[0]), b._new(x[1]))
        res.last_solve_stats = b.last_solve_stats = {"iterations": gstats["matvecs"], "outer": gstats["outer"]}
        if conv != 0:
            raise UserWarning("Warning:: Iterative solver is not converged ")
        return res

    @staticmethod
    def _solve_complex_block(H, bs, z, o, reverseGF):
        """The contour solves (z*I - H) x_i = b_i of ONE contour point for all right-hand sides in lock step
        (feast.py:198-200 runs them one after the other on the same operator and shift): each is the complex GCROT of
        ``_solve_complex``, unchanged, but their operator applications are collected and run as block products, four
        complex operands per pass over the operator (``gcrotmk_device_block``, ``hipeig_spmm_shift_pairs``).  Results,
        ``last_solve_stats`` and the exception on non-convergence are those of the one-by-one solves."""
        from .gcrotmk import gcrotmk_device_block
        ctx, n = bs[0].ctx, len(bs[0])
        H.honour_reduction_option(bs[0].options)
        rhs = []
        for b in bs:
            zero = ctx.alloc(n)
            _lib.call("hipeig_vec_fill", ctx.handle, zero.ptr, n, 0.0)
            rhs.append((b._buf, zero))

        def block_matvec(vs):
            return H.apply_shifted_pairs(z, vs, reverse=reverseGF)

        sols = gcrotmk_device_block(ctx, block_matvec, rhs, n, rtol=float(o["linear_tol"]), atol=float(o["linear_atol"]),
                                    maxiter=int(o["linearIter"]), complex_pairs=True,
                                    cols_per_pass=int(o.get("arnoldiColumnsPerPass", 1)))
        out, failed = [], False
        for b, (x, conv, gstats) in zip(bs, sols):
            res = HipComplexVector(b._new(x[0]), b._new(x[1]))
            res.last_solve_stats = b.last_solve_stats = {"iterations": gstats["matvecs"], "outer": gstats["outer"]}
            failed = failed or conv != 0
            out.append(res)
        if failed:
            raise UserWarning("Warning:: Iterative solver is not converged ")
        return out

    @staticmethod
    def _multi_dot(vectors, x):
        v0 = vectors[0]
        out = np.empty(len(vectors), dtype=np.float64)
        tab, keep = _ptr_table([v._buf for v in vectors])
        _lib.call("hipeig_multi_dot", v0.ctx.handle, v0._buf.n, len(vectors), tab, x._buf.ptr,
                  out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    @staticmethod
    def overlapMatrix(vectors):
        m = len(vectors)
        v0 = vectors[0]
        S = np.empty((m, m), dtype=np.float64)
        tab, keep = _ptr_table([v._buf for v in vectors])
        _lib.call("hipeig_gram", v0.ctx.handle, v0._buf.n, m, tab, m, tab,
                  S.ctypes.data_as(C.POINTER(C.c_double)))
        return np.triu(S) + np.triu(S, 1).T            # upper triangle mirrored, numpyVector.py:199-202

    @staticmethod
    def matrixRepresentation(operator, vectors):
        if not isinstance(operator, HipCsrOperator):
            raise TypeError("HipVector.matrixRepresentation needs a HipCsrOperator (device-resident CSR)")
        m = len(vectors)
        v0 = vectors[0]
        operator.honour_reduction_option(v0.options)
        # all kets H y_j in one block product, then one Gram block <y_i, H y_j> on the matrix cores
        kets = operator.apply_block([v._buf for v in vectors])
        M = np.empty((m, m), dtype=np.float64)
        ta, keep1 = _ptr_table([v._buf for v in vectors])
        tb, keep2 = _ptr_table(kets)
        _lib.call("hipeig_gram", v0.ctx.handle, v0._buf.n, m, ta, m, tb, M.ctypes.data_as(C.POINTER(C.c_double)))
        return np.tril(M) + np.tril(M, -1).T           # lower triangle mirrored, numpyVector.py:186-189

    @staticmethod
    def extendOverlapMatrix(vectors, overlap):
        col = HipVector._multi_dot(vectors, vectors[-1])
        m = len(vectors)
        S = np.empty((m, m), dtype=np.float64)
        S[:m - 1, :m - 1] = overlap
        S[:, m - 1] = col
        S[m - 1, :] = col
        return S

    @staticmethod
    def extendMatrixRepresentation(operator, vectors, opMat):
        ket = vectors[-1].applyOp(operator)
        col = HipVector._multi_dot(vectors, ket)
        m = len(vectors)
        M = np.empty((m, m), dtype=np.float64)
        M[:m - 1, :m - 1] = opMat
        M[:, m - 1] = col
        M[m - 1, :] = col
        return M


class HipComplexVector(AbstractVector):
    """complex128 vector on the device: two real fp64 halves (``re``, ``im``), every operation built from
    the real kernels on the halves (the operator is real, so ``H (a + i b) = H a + i H b``).

    It is what ``HipVector(array)`` returns for a complex array - the counterpart of a ``NumpyVector`` with a
    complex dtype (numpyVector.py:25-28, 89-93: ``vdot`` conjugates ``self`` unless ``conjugate=False``) -,
    what ``HipVector.solve`` returns for a complex shift (feast.py:90) and what a complex scalar times a
    ``HipVector`` gives (feast.py:91-92).  Same constructor convention as the other backends:
    ``HipComplexVector(complex_ndarray, options)``; internally also ``HipComplexVector(re_vec, im_vec)``."""

    def __init__(self, re, im=None, ctx=None):
        if isinstance(re, HipVector):
            self.re, self.im = re, im
        else:
            host = np.asarray(re, dtype=np.complex128)
            opts = im if isinstance(im, dict) or im is None else None
            self.re = HipVector(np.ascontiguousarray(host.real), opts, ctx=ctx)
            self.im = HipVector(np.ascontiguousarray(host.imag), self.re.options, ctx=ctx)
        self.options = self.re.options
        self.ctx = self.re.ctx
        self.size = self.re.size
        self.shape = self.re.shape
        self.last_solve_stats = None

    @staticmethod
    def fromArray(template, array):
        return HipComplexVector(np.asarray(array, dtype=np.complex128), template.options, ctx=template.ctx)

    @staticmethod
    def _parts(v):
        """(re, im) halves of a complex or real device vector (im is None for a real one)."""
        return (v.re, v.im) if isinstance(v, HipComplexVector) else (v, None)

    # ---- properties ----------------------------------------------------------------
    @property
    def hasExactAddition(self):
        return True

    @property
    def dtype(self):
        return np.dtype(np.complex128)

    @property
    def maxD(self):
        return 0

    @property
    def array(self):
        return self.re.array + 1j * self.im.array

    def __len__(self):
        return len(self.re)

    # ---- arithmetic ------------------------------------------------------------------
    def _scaled(self, alpha):
        a = complex(alpha)
        re = HipVector.linearCombination([self.re, self.im], [a.real, -a.imag])
        im = HipVector.linearCombination([self.re, self.im], [a.imag, a.real])
        return HipComplexVector(re, im)

    def __mul__(self, other):
        return self._scaled(other)

    def __rmul__(self, other):
        return self._scaled(other)

    def __truediv__(self, other):
        return self._scaled(1.0 / complex(other))

    def __imul__(self, other):
        raise NotImplementedError          # numpyVector.py:66-67

    def __itruediv__(self, other):
        raise NotImplementedError          # numpyVector.py:69-70

    # ---- instance methods ------------------------------------------------------------
    def norm(self):
        return float(np.hypot(self.re.norm(), self.im.norm()))

    def normalize(self):
        nrm = self.norm()
        for half in (self.re, self.im):    # in place: aliases of this vector see the change (numpyVector.py:76-78)
            _lib.call("hipeig_divide", self.ctx.handle, half._buf.n, nrm, half._buf.ptr, half._buf.ptr)
        return self

    def real(self):
        return self.re.copy()

    def conjugate(self):
        return HipComplexVector(self.re.copy(), self.im * -1.0)

    def vdot(self, other, conjugate=True):
        """numpyVector.py:89-93: ``np.vdot`` (self conjugated) or the bilinear ``np.dot``."""
        br, bi = self._parts(other)
        rr, ir = self.re.vdot(br), self.im.vdot(br)
        ri, ii = (self.re.vdot(bi), self.im.vdot(bi)) if bi is not None else (0.0, 0.0)
        if conjugate:
            return complex(rr + ii, ri - ir)
        return complex(rr - ii, ri + ir)

    def copy(self):
        return HipComplexVector(self.re.copy(), self.im.copy())

    def applyOp(self, other):
        if not isinstance(other, HipCsrOperator):
            raise TypeError("HipComplexVector.applyOp needs a HipCsrOperator (device-resident CSR)")
        out_r, out_i = self.re.ctx.alloc(other.nrows), self.re.ctx.alloc(other.nrows)
        other.apply_pair(self.re._buf, self.im._buf, out_r, out_i)
        return HipComplexVector(self.re._new(out_r), self.im._new(out_i))

    def compress(self):
        return self

    # ---- static hooks ------------------------------------------------------------------
    @staticmethod
    def linearCombination(vectors, coeffs):
        assert len(vectors) == len(coeffs)
        vs_re, cs_re, vs_im, cs_im = [], [], [], []
        for v, c in zip(vectors, coeffs):
            c = complex(c)
            vr, vi = HipComplexVector._parts(v)
            vs_re.append(vr); cs_re.append(c.real)          # Re: cr vr - ci vi
            vs_im.append(vr); cs_im.append(c.imag)          # Im: ci vr + cr vi
            if vi is not None:
                vs_re.append(vi); cs_re.append(-c.imag)
                vs_im.append(vi); cs_im.append(c.real)
        return HipComplexVector(HipVector.linearCombination(vs_re, cs_re), HipVector.linearCombination(vs_im, cs_im))

    @staticmethod
    def orthogonalize_against_set(x, qs, lindep=LINDEP_DEFAULT_VALUE):
        """The reference's sweep, numpyVector.py:121-145, with its bilinear products."""
        for q in qs:
            t1 = x.vdot(q, conjugate=False)
            t2 = q.vdot(q, conjugate=False)
            x = HipComplexVector.linearCombination([x, q * (t1 / t2)], [1.0, -1.0])
        ip = np.complex128(x.vdot(x, conjugate=False))
        if ip > lindep:                                       # NumPy orders complex numbers lexicographically, as in the reference
            return x / np.sqrt(ip)
        return None

    @staticmethod
    def solve(H, b, sigma, x0=None, opType="her", reverseGF=False):
        """(sigma I - H) x = b for a complex right-hand side.  Real shift with MINRES: the real operator acts on
        the halves separately, so the two real systems go through ``solveBlock`` - which solves TWO right-hand sides one
        after the other (the lock-step block kernel pays from three on, ``BLOCK_SOLVE_MIN``); everything else is complex
        GCROT on (re, im) pairs, like a complex shift."""
        if not isinstance(H, HipCsrOperator):
            raise TypeError("HipComplexVector.solve needs a HipCsrOperator (device-resident CSR)")
        o = b.options["linearSystemArgs"]
        if o["linearSolver"] == "pardiso":
            return HipVector._solve_exact_small(H, b, sigma, reverseGF)
        is_complex_shift = isinstance(sigma, complex) or np.iscomplexobj(sigma)
        if x0 is not None and not isinstance(x0, (HipVector, HipComplexVector)):
            x0 = HipVector(np.asarray(x0), ctx=b.ctx)
        if o["linearSolver"] == "minres" and not is_complex_shift:
            if x0 is None:
                xr, xi = HipVector.solveBlock(H, [b.re, b.im], sigma, reverseGF=reverseGF)
            else:                                       # the real operator acts on the halves separately: so does the guess
                g_re, g_im = (x0.re, x0.im) if isinstance(x0, HipComplexVector) else (x0, x0 * 0.0)
                xr = HipVector.solve(H, b.re, sigma, g_re, reverseGF=reverseGF)
                xi = HipVector.solve(H, b.im, sigma, g_im, reverseGF=reverseGF)
            res = HipComplexVector(xr, xi)
            res.last_solve_stats = b.last_solve_stats = {"re": xr.last_solve_stats, "im": xi.last_solve_stats,
                                                         "iterations": max(xr.last_solve_stats["iterations"],
                                                                           xi.last_solve_stats["iterations"])}
            return res
        return HipVector._solve_complex(H, b, complex(sigma), o, reverseGF, x0)

    @staticmethod
    def _gram(A, B):
        """A^H B for lists of (possibly complex) vectors: four real Gram blocks on the matrix cores."""
        ar = [HipComplexVector._parts(v)[0] for v in A]
        ai = [HipComplexVector._parts(v)[1] for v in A]
        br = [HipComplexVector._parts(v)[0] for v in B]
        bi = [HipComplexVector._parts(v)[1] for v in B]

        def g(X, Y):
            if any(v is None for v in X) or any(v is None for v in Y):
                return np.zeros((len(X), len(Y)))
            out = np.empty((len(X), len(Y)))
            tx, k1 = _ptr_table([v._buf for v in X])
            ty, k2 = _ptr_table([v._buf for v in Y])
            _lib.call("hipeig_gram", X[0].ctx.handle, X[0]._buf.n, len(X), tx, len(Y), ty, out.ctypes.data_as(C.POINTER(C.c_double)))
            return out
        return (g(ar, br) + g(ai, bi)) + 1j * (g(ar, bi) - g(ai, br))

    @staticmethod
    def overlapMatrix(vectors):
        S = HipComplexVector._gram(vectors, vectors)
        return np.triu(S) + np.triu(S, 1).conj().T           # upper triangle mirrored, numpyVector.py:199-202

    @staticmethod
    def matrixRepresentation(operator, vectors):
        kets = [v.applyOp(operator) for v in vectors]
        M = HipComplexVector._gram(vectors, kets)
        return np.tril(M) + np.tril(M, -1).conj().T           # lower triangle mirrored, numpyVector.py:186-189

    @staticmethod
    def extendOverlapMatrix(vectors, overlap):
        col = HipComplexVector._gram(vectors, vectors[-1:])[:, 0]
        m = len(vectors)
        S = np.empty((m, m), dtype=np.complex128)
        S[:m - 1, :m - 1] = overlap
        S[:, m - 1] = col
        S[m - 1, :m - 1] = col[:-1].conj()
        return S

    @staticmethod
    def extendMatrixRepresentation(operator, vectors, opMat):
        col = HipComplexVector._gram(vectors, [vectors[-1].applyOp(operator)])[:, 0]
        m = len(vectors)
        M = np.empty((m, m), dtype=np.complex128)
        M[:m - 1, :m - 1] = opMat
        M[:, m - 1] = col
        M[m - 1, :m - 1] = col[:-1].conj()
        return M

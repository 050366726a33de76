"""Round-trip latency of the host-scalar calls at a length where the kernel itself is nothing: dot (mapped result + stream
synchronise), scale (launch only, no wait), and the split Arnoldi step's begin/end (event synchronise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
import eigensolvers_amd as ea
from eigensolvers_amd import _lib
from eigensolvers_amd.gcrotmk import _PairOps
ctx = ea.HipContext.default()
n = 1024
x = ea.HipVector(np.random.default_rng(0).standard_normal(n))
y = ctx.alloc(n)
out = C.c_double()
for name, fn in (("dot", lambda: _lib.call("hipeig_dot", ctx.handle, n, x._buf.ptr, x._buf.ptr, C.byref(out))),
                 ("scale (no wait)", lambda: _lib.call("hipeig_scale", ctx.handle, n, 1.0, x._buf.ptr, y.ptr))):
    for _ in range(200):
        fn()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(5000):
        fn()
    ctx.synchronize()
    print("%-18s %.2f us per call" % (name, (time.perf_counter() - t) / 5000 * 1e6))
ops = _PairOps(ctx, n)
w = (ea.HipVector(np.ones(n))._buf, ea.HipVector(np.ones(n))._buf)
for _ in range(200):
    ops.arnoldi_begin([], w, 0); ops.arnoldi_end(0, 0)
t = time.perf_counter()
for _ in range(5000):
    ops.arnoldi_begin([], w, 0); ops.arnoldi_end(0, 0)
print("%-18s %.2f us per call" % ("arnoldi begin+end", (time.perf_counter() - t) / 5000 * 1e6))

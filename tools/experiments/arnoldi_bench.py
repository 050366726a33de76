"""Time of one Arnoldi step (hipeig_pair_arnoldi_step_p) against m complex columns: the sequential one-column sweep and the
blocked four-column form.  python tools/experiments/arnoldi_bench.py N m [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
from eigensolvers_amd.gcrotmk import _PairOps, _Ops
N, m = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = ea.HipContext.default()
rng = np.random.default_rng(0)
V = [(ea.HipVector(rng.standard_normal(N) / np.sqrt(2 * N))._buf, ea.HipVector(rng.standard_normal(N) / np.sqrt(2 * N))._buf) for _ in range(m)]
w0 = (rng.standard_normal(N), rng.standard_normal(N))
t_go = float(os.environ.get("ARN_START_AT", "0"))      # concurrent copies: all start their timed loops at this wall-clock time
for cols in ((4,) if t_go else (1, 4, 1, 4)):
    ops = _PairOps(ctx, N, cols)
    ts = []
    while time.time() < t_go:
        time.sleep(0.001)
    for r in range(reps):
        w = (ea.HipVector(w0[0].copy())._buf, ea.HipVector(w0[1].copy())._buf)
        ctx.synchronize()
        t = time.perf_counter()
        nb, h, na = ops.arnoldi_step(V, w)
        ts.append(time.perf_counter() - t)
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"N {N} m {m} cols {cols}: {med * 1e3:.3f} ms per step = {med / max(m, 1) * 1e6:.2f} us per column; h[0] {h[0] if m else 0:.6e} na {na:.12e}", flush=True)

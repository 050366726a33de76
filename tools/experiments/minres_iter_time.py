"""MINRES time per iteration with KD riding on the next sweep (default) and as its own kernel (HIPEIG_MINRES_FUSE_KD=0).
python tools/experiments/minres_iter_time.py N nnz_row"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
N, nnz_row = int(sys.argv[1]), int(sys.argv[2])
ctx = ea.HipContext.default()
H = ea.HipCsrOperator.generate(N, nnz_row, seed=7)
x = np.random.default_rng(0).standard_normal(N); x /= np.linalg.norm(x)
res = {}
for mode in ("1", "0", "1", "0"):
    os.environ["HIPEIG_MINRES_FUSE_KD"] = mode
    X = ea.HipVector(x.copy(), {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 4000, "linear_tol": 1e-10}})
    ctx.synchronize()
    t = time.perf_counter()
    W = ea.HipVector.solve(H, X, 0.02)
    ctx.synchronize()
    dt = time.perf_counter() - t
    st = W.last_solve_stats
    print(f"N {N} fuse_kd {mode}: {st['iterations']} its, istop {st['istop']}, {dt / st['iterations'] * 1e3:.4f} ms/it, |x| {W.norm():.15e}", flush=True)

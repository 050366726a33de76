"""Column splits (full-height row blocks shared by `csplit` workgroups) against the plain TCOO-W sweep at sizes below the
N = 4e6 switch point: product time and MINRES iteration time.  python tools/experiments/csplit_sweep.py N nnz_row csplit..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
N, nnz_row = int(sys.argv[1]), int(sys.argv[2])
ctx = ea.HipContext.default()
x = np.random.default_rng(0).standard_normal(N); x /= np.linalg.norm(x)
for cs in [int(v) for v in sys.argv[3:]]:
    os.environ["HIPEIG_TCOOW_CSPLIT"] = str(cs)
    H = ea.HipCsrOperator.generate(N, nnz_row, seed=7)
    H.set_variant(4)
    X = ea.HipVector(x.copy(), {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 300, "linear_tol": 1e-30}})
    y = ctx.alloc(N)
    H.apply(X._buf, y); ctx.synchronize()
    ctx.timer_start()
    for _ in range(50):
        H.apply(X._buf, y)
    ms = ctx.timer_stop() / 50
    t = time.perf_counter()
    try:
        ea.HipVector.solve(H, X, 0.02)
    except UserWarning:
        pass
    its = X.last_solve_stats["iterations"]
    dt = time.perf_counter() - t
    print(f"N {N} csplit {cs}: launches/product {H.launches_per_apply()} product {ms:.4f} ms  MINRES {its} its {dt / its * 1e3:.4f} ms/it", flush=True)
    del H

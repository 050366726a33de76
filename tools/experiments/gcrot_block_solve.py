"""The 16 contour solves of ONE contour point in lock step (config #5's unit of work) at N; run under rocprofv3
--kernel-trace --stats to see how the time divides.  python tools/experiments/gcrot_block_solve.py [N [cols_per_pass [nrhs]]]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nrhs = int(sys.argv[3]) if len(sys.argv) > 3 else 16
H = ea.HipCsrOperator.generate(N, 32 if N <= 2_000_000 else 64, seed=7)
opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 4000, "linear_tol": 1e-5, "linear_atol": 1e-7,
                            "arnoldiColumnsPerPass": cols}}
rng = np.random.default_rng(9)
bs = [ea.HipVector(rng.standard_normal(N), opt) for _ in range(nrhs)]
for b in bs:
    b.normalize()
z = 0.0 + 0.21 * np.exp(1j * np.pi * 0.03)          # a contour point close to the real axis (the expensive ones)
warnings.simplefilter("ignore")
ctx = ea.HipContext.default()
ctx.synchronize()
t = time.perf_counter()
ws = ea.HipVector.solveBlock(H, bs, complex(z))
ctx.synchronize()
dt = time.perf_counter() - t
its = [w.last_solve_stats["iterations"] for w in ws]
print(f"N {N} cols {cols}: {nrhs} right-hand sides, products {its}, {dt:.3f} s = {dt / sum(its) * 1e3:.4f} ms per product and right-hand side")

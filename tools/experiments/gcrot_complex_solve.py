"""One complex contour solve (z*I - H) x = b with GCROT at N (config #5's inner kernel mix); run under
rocprofv3 --kernel-trace --stats to see how an inner step divides between the product and the Arnoldi sweep.
python tools/experiments/gcrot_complex_solve.py [N]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
H = ea.HipCsrOperator.generate(N, 32 if N <= 2_000_000 else 64, seed=7)
opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 4000, "linear_tol": 1e-5, "linear_atol": 1e-7}}
b = ea.HipVector(np.random.default_rng(9).standard_normal(N), opt)
b.normalize()
z = 0.0 + 0.21 * np.exp(1j * np.pi * 0.03)          # a contour point close to the real axis (the expensive ones)
warnings.simplefilter("ignore")
t = time.perf_counter()
w = ea.HipVector.solve(H, b, complex(z))
ea.HipContext.default().synchronize()
dt = time.perf_counter() - t
st = w.last_solve_stats
print(f"N {N}: {st['iterations']} products, {st['outer']} outer cycles, {dt:.3f} s, {dt / st['iterations'] * 1e3:.3f} ms per inner step")

"""BASELINE config #5 shape at N = 1e6 run to the reference's stopping rule; writes a JSON summary."""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as la
import eigensolvers_amd as ea
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tol, econv, maxit, m0 = 1e-5, 1e-4, 12, 16
H = ea.HipCsrOperator.generate(N, 32 if N <= 2_000_000 else 64, seed=7)
Q = la.qr(np.random.default_rng(9).standard_normal((N, m0)), mode="economic")[0]
opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 4000, "linear_tol": tol, "linear_atol": tol * 1e-2}}
t = time.time()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ev, Y, st = ea.feastDiagonalization(H, [ea.HipVector(Q[:, i].copy(), opt) for i in range(m0)], 16, "legendre", -0.21, 0.21, econv, maxit, writeOut=False)
dt = time.time() - t
res = ea.true_residual_norms(H, ev, Y, len(Y))
print(json.dumps({"config": "FEAST, window [-0.21, 0.21], nc = 16 (8 half-contour points), m0 = 16, gcrotmk rtol 1e-5, eConv 1e-4",
                  "N": N, "nnz": int(H.nnz), "outerIter": int(st["outerIter"]), "residual": float(st["residual"]), "converged": bool(st["residual"] < econv),
                  "seconds": round(dt, 1), "seconds_per_feast_iteration": round(dt / (st["outerIter"] + 1), 1),
                  "eigenvalues_in_window": np.sort(ev[(ev > -0.21) & (ev < 0.21)]).tolist(), "true_residual_norms": res.tolist()}, indent=1))

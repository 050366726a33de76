"""Which sweep layouts give bit-identical MINRES iterates with KD in the epilogue and KD as its own kernel?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
from eigensolvers_amd.generators import gapped_csr_host, guess_vector
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
its = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
Hh = gapped_csr_host(n, 16, seed=7)
b = guess_vector(n, 1); b = b / np.linalg.norm(b)
def solve(variant, fuse):
    os.environ["HIPEIG_MINRES_FUSE_KD"] = fuse
    H = ea.HipCsrOperator.from_scipy(Hh); H.set_variant(variant)
    try:
        W = ea.HipVector.solve(H, ea.HipVector(b.copy(), {"linearSystemArgs": {"linearSolver": "minres", "linearIter": its, "linear_tol": 1e-8}}), 0.02)
    except UserWarning:
        return None, None
    return W.array, W.last_solve_stats
for v in (1, 2, 3, 5):
    a, sa = solve(v, "0"); b2, sb = solve(v, "0"); c, sc = solve(v, "1"); d, sd = solve(v, "1")
    print(f"variant {v}: its {sa['iterations']} {sc['iterations']}  unfused==unfused {np.array_equal(a, b2)}  fused==fused {np.array_equal(c, d)}  unfused==fused {np.array_equal(a, c)}  maxdiff {np.max(np.abs(a - c)):.3e}", flush=True)

"""BASELINE config #4 as a PARTITIONED run on one GPU: the N = 1e7, 64 nnz/row operator row-partitioned over 8 ranks (8
contexts of this process joined by the library's loopback collectives: the chunk-major exchange, column splits, two
collectives per MINRES iteration - everything of an 8-GPU run except the transport), inexact Lanczos to convergence, and
the same run on the whole operator in a ninth context.  JSON on stdout.
python tools/experiments/config4_loopback8.py [N [P]]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
from eigensolvers_amd.distributed import LoopbackGroup, row_range
from eigensolvers_amd.generators import guess_vector
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
opts = lambda: {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 4000, "linear_tol": 1e-10, "linear_atol": 1e-12}}
t0 = time.time()
H = ea.HipCsrOperator.generate(N, 64, seed=7)
ev1, Y1, st1 = ea.inexactLanczosDiagonalization(H, ea.HipVector(guess_vector(N, 1).copy(), opts()), 0.02, 8, 4, 1e-10, writeOut=False)
res1 = float(ea.true_residual_norms(H, ev1, Y1, 1)[0])
t_single = time.time() - t0
del H, Y1
grp = LoopbackGroup(P)

def body(rank, ctx):
    b, e = row_range(N, P, rank)
    Hr = ea.HipCsrOperator.generate(N, 64, seed=7, row_begin=b, row_end=e, ctx=ctx)
    v0 = ea.HipVector(guess_vector(N, 1, b, e).copy(), opts(), ctx=ctx)
    t = time.time()
    ev, Y, st = ea.inexactLanczosDiagonalization(Hr, v0, 0.02, 8, 4, 1e-10, writeOut=False)
    dt = time.time() - t
    res = float(ea.true_residual_norms(Hr, ev, Y, 1)[0])
    return {"ev0": float(ev[0]), "cumIter": int(st["cumIter"]), "converged": bool(st["isConverged"]), "residual": float(st["residual"]),
            "true_residual": res, "seconds": dt, "layout": Hr.layout_info(), "minres_last": Y[0].last_solve_stats}

try:
    out = grp.run(body)
finally:
    grp.close()
print(json.dumps({"config": f"BASELINE #4: N = {N}, 64 nnz/row, row-partitioned over {P} loopback ranks on ONE GPU, inexact Lanczos L = 8, eConv 1e-10, minres rtol 1e-10",
                  "single_context": {"ev0": float(ev1[0]), "cumIter": int(st1["cumIter"]), "true_residual": res1, "seconds_incl_build": round(t_single, 2)},
                  "ranks": out,
                  "all_ranks_identical_ev0": len({o["ev0"] for o in out}) == 1,
                  "rel_diff_to_single": abs(out[0]["ev0"] - float(ev1[0])) / abs(float(ev1[0]))}, indent=1))

"""FEAST with the 8 contour points dealt to P in-process replicas on ONE GPU (distributed.LoopbackGroup +
ContourReplicas): overlaps the launch-bound GCROT solves of different contour points.  python ... N P"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as la
import eigensolvers_amd as ea
from eigensolvers_amd.distributed import LoopbackGroup, ContourReplicas
N = int(sys.argv[1]); P = int(sys.argv[2]); m0 = 16
Q = la.qr(np.random.default_rng(9).standard_normal((N, m0)), mode="economic")[0]
opt = lambda: {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 4000, "linear_tol": 1e-5, "linear_atol": 1e-7}}

def run(ctx, comm):
    H = ea.HipCsrOperator.generate(N, 32, seed=7, ctx=ctx)
    Y = [ea.HipVector(Q[:, i].copy(), opt(), ctx=ctx) for i in range(m0)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Yf, st = ea.feastDiagonalization(H, Y, 16, "legendre", -0.21, 0.21, 1e-4, 12, writeOut=False, contourComm=comm)
    return np.sort(ev), st["outerIter"], st["residual"]

t = time.time()
if P == 1:
    ev, it, res = run(ea.HipContext.default(), None)
else:
    grp = LoopbackGroup(P)
    try:
        out = grp.run(lambda rank, ctx: run(ctx, ContourReplicas(ctx)))
    finally:
        grp.close()
    ev, it, res = out[0]
print("N", N, "replicas", P, "outer", it, "res", res, "t %.1f" % (time.time() - t), "ev[8]", ev[8])

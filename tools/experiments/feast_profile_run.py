"""BASELINE config #5 shape run to the reference's stopping rule (or for argv[2] FEAST iterations); JSON summary on stdout.
python tools/experiments/feast_profile_run.py [N [maxit [arnoldi columns per pass: 1 | 4 [lock-step block solves: 1 | 0 [gcrotmk rtol [eConv [window half-width [m0]]]]]]]]
The summary file (one line per FEAST iteration: eigenvalues, eigenvalue-change residual, seconds) goes to
gpurun_out/feast_summary_N<N>.out as the run proceeds, so a run that is cut off still leaves its trajectory."""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as la
import eigensolvers_amd as ea
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tol = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-5
econv = float(sys.argv[6]) if len(sys.argv) > 6 else 1e-4
maxit, m0 = (int(sys.argv[2]) if len(sys.argv) > 2 else 12), (int(sys.argv[8]) if len(sys.argv) > 8 else 16)
hw = float(sys.argv[7]) if len(sys.argv) > 7 else 0.21
H = ea.HipCsrOperator.generate(N, 32 if N <= 2_000_000 else 64, seed=7)
Q = la.qr(np.random.default_rng(9).standard_normal((N, m0)), mode="economic")[0]
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 1
block = (int(sys.argv[4]) != 0) if len(sys.argv) > 4 else True
opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 4000, "linear_tol": tol, "linear_atol": tol * 1e-2,
                            "arnoldiColumnsPerPass": cols}, "blockSolve": block}
import threading
def _heartbeat(t0=time.time()):                 # a line a minute: a silent long run is taken for a hung one on the GPU box
    while True:
        time.sleep(60)
        print("... %.0f s" % (time.time() - t0), file=sys.stderr, flush=True)
threading.Thread(target=_heartbeat, daemon=True).start()
t = time.time()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    ev, Y, st = ea.feastDiagonalization(H, [ea.HipVector(Q[:, i].copy(), opt) for i in range(m0)], 16, "legendre", -hw, hw, econv, maxit,
                                        writeOut=True, summaryFileName=os.path.join(out_dir, f"feast_summary_N{N}.out"))
dt = time.time() - t
res = ea.true_residual_norms(H, ev, Y, len(Y))
print(json.dumps({"config": f"FEAST, window [{-hw:g}, {hw:g}], nc = 16 (8 half-contour points), m0 = {m0}, gcrotmk rtol {tol:g}, eConv {econv:g}, maxit {maxit}",
                  "arnoldi_columns_per_pass": cols, "lock_step_block_solves": block, "N": N, "nnz": int(H.nnz), "outerIter": int(st["outerIter"]), "residual": (None if st["residual"] is None else float(st["residual"])), "converged": bool(st["residual"] is not None and st["residual"] < econv),
                  "seconds": round(dt, 1), "seconds_per_feast_iteration": round(dt / (st["outerIter"] + 1), 1),
                  "eigenvalues_in_window": np.sort(ev[(ev > -hw) & (ev < hw)]).tolist(), "true_residual_norms": res.tolist()}, indent=1))

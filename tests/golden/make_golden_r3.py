"""Round-3 golden values: the REAL reference run at BASELINE sizes (same harness as make_golden.py).

* config2_n1e6.json   config #2 - gapped CSR N = 1e6, 32 nnz/row, seed 7, single-vector inexact Lanczos
                      (inexact_Lanczos.py:229-443 through numpyVector.py:147-178, scipy minres rtol 1e-10),
                      sigma = 0.02, L = 8, maxit = 4, eConv 1e-12: exactly the run of
                      tests/test_gpu_fullsize.py::single_1e6.  Stored: ev[0], cumIter, residual, isConverged,
                      the true residual norm of the returned Ritz pair and the reference's wall time.
* config3_n1e6.json   config #3 - the same operator, block of 8 (the BLOCK8_* parameters of test_gpu_fullsize.py).
* config4_n<N>.json   a reduced config-#4 instance: 64 nnz/row at the largest N the container's RAM holds.
* config5_feast_n<N>.json   config #5's FEAST recipe (feast.py:126-244: window [-0.21, 0.21], nc = 16 -> 8 half-contour points,
                      m0 = 16, gcrotmk rtol 1e-3) at N = 1e5 - what a few hours of CPU allow (`feast:<N>`).

Inputs come from eigensolvers_amd.generators by seed; only a few hundred bytes of outputs are stored.
Runs only in the build container (needs /root/reference), CPU only:

    python tests/golden/make_golden_r3.py config2 [config3] [config4:<N>]
"""
import json
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

CONFIG2 = dict(N=1_000_000, nnz_row=32, seed=7, sigma=0.02, L=8, maxit=4, eConv=1e-12, linear_tol=1e-10,
               linearIter=4000, guess_seed=1)
CONFIG3 = dict(N=1_000_000, nnz_row=32, seed=7, sigma=0.02, L=12, maxit=2, eConv=1e-12, linear_tol=1e-11,
               linearIter=4000, guess_seed=5, nBlock=8)


def _opts(p):
    return {"linearSystemArgs": {"linearSolver": "minres", "linearIter": p["linearIter"], "linear_tol": p["linear_tol"],
                                 "linear_atol": p["linear_tol"] * 1e-2}}


def _true_residuals(H, ev, Y, k):
    out = []
    for i in range(k):
        y = Y[i].array
        out.append(float(np.linalg.norm(H @ y - ev[i] * y)))
    return out


def run_single(p, name):
    from make_golden import import_reference
    from eigensolvers_amd.generators import gapped_csr_host, guess_vector
    iL, NumpyVector, uf = import_reference()
    t0 = time.time()
    H = gapped_csr_host(p["N"], p["nnz_row"], seed=p["seed"])
    print(f"{name}: operator built in {time.time() - t0:.1f} s, nnz = {H.nnz}", flush=True)
    v0 = NumpyVector(guess_vector(p["N"], p["guess_seed"]).copy(), _opts(p))
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = iL.inexactLanczosDiagonalization(H, v0, p["sigma"], p["L"], p["maxit"], p["eConv"], writeOut=False,
                                                     saveTNSsEachIteration=False)
    wall = time.time() - t0
    out = dict(p)
    out.update(ev0=float(np.real(ev[0])), ev=[float(np.real(e)) for e in ev], cumIter=int(st["cumIter"]),
               residual=float(st["residual"]), isConverged=bool(st["isConverged"]), nvec=len(Y),
               true_residual=_true_residuals(H, ev, Y, 1), reference_wall_s=wall, nnz=int(H.nnz),
               numpy=np.__version__, scipy=__import__("scipy").__version__)
    json.dump(out, open(os.path.join(HERE, name), "w"), indent=1)
    print(name, out["ev0"], out["cumIter"], out["residual"], out["isConverged"], f"{wall:.0f} s", flush=True)


def run_block(p, name):
    import scipy.linalg as la
    from make_golden import import_reference
    from eigensolvers_amd.generators import gapped_csr_host
    iL, NumpyVector, uf = import_reference()
    H = gapped_csr_host(p["N"], p["nnz_row"], seed=p["seed"])
    Q = la.qr(np.random.default_rng(p["guess_seed"]).standard_normal((p["N"], p["nBlock"])), mode="economic")[0]
    v0 = [NumpyVector(Q[:, i].copy(), _opts(p)) for i in range(p["nBlock"])]
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = iL.inexactLanczosDiagonalization(H, v0, p["sigma"], p["L"], p["maxit"], p["eConv"], writeOut=False,
                                                     saveTNSsEachIteration=False)
    wall = time.time() - t0
    out = dict(p)
    k = p["nBlock"]
    out.update(ev=[float(np.real(e)) for e in ev], cumIter=int(st["cumIter"]), residual=float(st["residual"]),
               isConverged=bool(st["isConverged"]), nvec=len(Y), true_residual=_true_residuals(H, ev, Y, k),
               reference_wall_s=wall, nnz=int(H.nnz), numpy=np.__version__, scipy=__import__("scipy").__version__)
    json.dump(out, open(os.path.join(HERE, name), "w"), indent=1)
    print(name, np.sort(ev[:k]), out["cumIter"], out["isConverged"], f"{wall:.0f} s", flush=True)


FEAST5 = dict(nnz_row=32, seed=7, m0=16, nc=16, eMin=-0.21, eMax=0.21, eConv=1e-4, maxit=12, linear_tol=1e-3, linear_atol=1e-5,
              linearIter=4000, guess_seed=9)


def run_feast(N, name):
    """Config #5's recipe (window, 8 half-contour points, m0 = 16, gcrotmk) through the reference's feast.py:126-244."""
    import scipy.linalg as la
    from make_golden import import_reference
    from eigensolvers_amd.generators import gapped_csr_host
    import_reference()
    import feast as rf
    from numpyVector import NumpyVector
    p = dict(FEAST5, N=N)
    H = gapped_csr_host(N, p["nnz_row"], seed=p["seed"])
    Q = la.qr(np.random.default_rng(p["guess_seed"]).standard_normal((N, p["m0"])), mode="economic")[0]
    opt = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": p["linearIter"], "linear_tol": p["linear_tol"],
                                "linear_atol": p["linear_atol"]}}
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, Y, st = rf.feastDiagonalization(H, [NumpyVector(Q[:, i].copy(), opt) for i in range(p["m0"])], p["nc"], "legendre",
                                            p["eMin"], p["eMax"], p["eConv"], p["maxit"], writeOut=False)
    wall = time.time() - t0
    out = dict(p)
    out.update(ev=[float(np.real(e)) for e in ev], outerIter=int(st["outerIter"]), residual=float(st["residual"]), nvec=len(Y),
               true_residual=_true_residuals(H, ev, Y, len(Y)), reference_wall_s=wall, nnz=int(H.nnz), numpy=np.__version__,
               scipy=__import__("scipy").__version__)
    json.dump(out, open(os.path.join(HERE, name), "w"), indent=1)
    print(name, out["outerIter"], out["residual"], f"{wall:.0f} s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["config2"]
    for w in what:
        if w == "config2":
            run_single(CONFIG2, "config2_n1e6.json")
        elif w == "config3":
            run_block(CONFIG3, "config3_n1e6.json")
        elif w.startswith("feast:"):
            n = int(float(w.split(":")[1]))
            run_feast(n, f"config5_feast_n{n}.json")
        elif w.startswith("config4:"):
            n = int(float(w.split(":")[1]))
            p = dict(CONFIG2, N=n, nnz_row=64, eConv=1e-10)
            run_single(p, f"config4_n{n}.json")

#!/usr/bin/env python3
"""Block path timings (BASELINE config #3): the k = 8 block product in both kernel forms against 8
single products, and the lock-step block MINRES against 8 single solves on the same right-hand sides.

    python tools/block_bench.py [--n 1000000 --nnz-row 32 --k 8 --rtol 1e-10] > profiles/rNN_block.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nnz-row", type=int, default=32)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-solve", action="store_true")
    ap.add_argument("--variants", default="1,2")
    a = ap.parse_args()
    import numpy as np
    import scipy.linalg as la
    import eigensolvers_amd as ea
    ctx = ea.HipContext.default()
    H = ea.HipCsrOperator.generate(a.n, a.nnz_row, seed=7)
    rng = np.random.default_rng(5)
    Q = rng.standard_normal((a.n, a.k))
    Q /= np.linalg.norm(Q, axis=0)
    opts = lambda: {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 4000, "linear_tol": a.rtol}}
    X = [ea.HipVector(Q[:, j].copy(), opts()) for j in range(a.k)]
    out = {"N": a.n, "nnz": int(H.nnz), "k": a.k, "device": ctx.device_info()["name"]}

    def timed(fn, reps):
        fn()
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps

    y = ctx.alloc(a.n)
    out["spmv_ms"] = timed(lambda: H.apply(X[0]._buf, y), a.reps)
    out["k_single_products_ms"] = out["spmv_ms"] * a.k
    ref = [ea.HipVector(X[j]._buf).applyOp(H) for j in range(a.k)]
    alg = H.nnz * 12 + 4 * (a.n + 1) + 2 * 8 * a.n * a.k
    for bv in [int(v) for v in a.variants.split(",")]:
        H.set_block_variant(bv)
        t0 = time.time()
        Y = H.apply_block([x._buf for x in X])
        ctx.synchronize()
        build = time.time() - t0
        ms = timed(lambda: H.apply_block([x._buf for x in X]), a.reps)
        err = max(ea.HipVector.linearCombination([ea.HipVector(Y[j]), ref[j]], [1.0, -1.0]).norm() / ref[j].norm() for j in range(a.k))
        info = H.block_info()
        out[f"spmm_{info['variant']}"] = {"ms_incl_pack_unpack": ms, "first_call_s": build, "max_rel_diff_vs_single": err,
                                          "algorithmic_GBs": alg / ms / 1e6, "layout": info}
    if not a.no_solve:
        H.set_block_variant(0)
        t = time.time()
        W = ea.HipVector.solveBlock(H, X, 0.02)
        ctx.synchronize()
        t_block = time.time() - t
        t = time.time()
        W = ea.HipVector.solveBlock(H, X, 0.02)
        ctx.synchronize()
        t_block = min(t_block, time.time() - t)
        its = [w.last_solve_stats["iterations"] for w in W]
        t = time.time()
        S = [ea.HipVector.solve(H, x, 0.02) for x in X]
        ctx.synchronize()
        t_single = time.time() - t
        its1 = [s.last_solve_stats["iterations"] for s in S]
        diff = max(ea.HipVector.linearCombination([W[j], S[j]], [1.0, -1.0]).norm() / S[j].norm() for j in range(a.k))
        out["block_solve"] = {"kernel": H.block_info()["variant"], "seconds": t_block, "iterations": its,
                              "solves_per_s": a.k / t_block, "ms_per_block_iteration": 1e3 * t_block / max(its)}
        out["single_solves"] = {"seconds": t_single, "iterations": its1, "solves_per_s": a.k / t_single,
                                "ms_per_iteration": 1e3 * t_single / sum(its1)}
        out["block_vs_single_speedup"] = t_single / t_block
        out["max_rel_diff_block_vs_single_solution"] = diff
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

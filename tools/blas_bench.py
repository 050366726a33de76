#!/usr/bin/env python3
"""Roofline fractions of the BLAS-1 and tall-skinny kernels at N = 1e7, through the public API.

Every case cycles through enough distinct vectors (>= 1.3 GB) that the 256 MiB Infinity Cache cannot
hold the operands from one repetition to the next, so the rates are HBM rates.  Timing: HIP events on
the library's compute stream (hipeig_timer_start/stop) around `reps` back-to-back calls.  Host-scalar
calls (dot, gram ...) synchronise on return; their launch + copy-back overhead is part of the number.

usage: python tools/blas_bench.py [out.json] [N]
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import eigensolvers_amd as ea  # noqa: E402

PEAK = 8.0e12


def main():
    args = [a for a in sys.argv[1:]]
    out = args[0] if args else os.path.join(REPO, "gpurun_out", "blas_bench.json")
    N = int(args[1]) if len(args) > 1 else 10_000_000
    ctx = ea.HipContext.default()
    rng = np.random.default_rng(0)
    host = rng.standard_normal(N)
    pool = [ea.HipVector(np.roll(host, 17 * i) * (1.0 + 0.01 * i)) for i in range(34)]   # 34 x 80 MB = 2.7 GB
    V = ea.HipVector
    rep = {"device": ctx.device_info()["name"], "N": N, "peak_GBps": PEAK / 1e9, "cases": []}

    def timed(name, algo_bytes, fn, reps=20):
        fn(0)
        ctx.synchronize()
        ctx.timer_start()
        for r in range(reps):
            fn(r + 1)
        ms = ctx.timer_stop() / reps
        gbps = algo_bytes / (ms * 1e-3) / 1e9
        rep["cases"].append({"kernel": name, "algorithmic_bytes": algo_bytes, "ms": round(ms, 4),
                             "GBps": round(gbps, 1), "frac": round(gbps * 1e9 / PEAK, 4)})
        print(f"{name:46s} {ms:8.3f} ms  {gbps:8.1f} GB/s  {100 * gbps * 1e9 / PEAK:5.1f} %", file=sys.stderr, flush=True)

    B = 8 * N
    pick = lambda r, k, stride=1: [pool[(r * k + i * stride) % len(pool)] for i in range(k)]
    timed("dot (vdot)", 2 * B, lambda r: pick(r, 2)[0].vdot(pick(r, 2)[1]))
    timed("nrm2 (norm)", B, lambda r: pool[r % len(pool)].norm())
    timed("scale (__mul__)", 2 * B, lambda r: pool[r % len(pool)] * 1.5)
    timed("lincomb k=2 (axpy shape)", 3 * B, lambda r: V.linearCombination(pick(r, 2), [1.0, -0.5]))
    for k in (4, 8, 16):
        cf = list(np.linspace(0.5, 1.5, k))
        timed(f"lincomb k={k}", (k + 1) * B, lambda r, k=k, cf=cf: V.linearCombination(pick(r, k), cf))
    for m in (4, 8, 16, 32):
        timed(f"multi_dot m={m} (extendOverlapMatrix col)", (m + 1) * B,
              lambda r, m=m: V._multi_dot(pick(r, m), pool[(r * m + m) % len(pool)]))
    for m in (8, 16, 32):
        timed(f"gram {m}x{m} (overlapMatrix, MFMA)", m * B, lambda r, m=m: V.overlapMatrix(pick(r, m)), reps=10)
    for m, k in ((16, 8), (32, 8)):
        C = rng.standard_normal((m, k))
        timed(f"lincomb_block {m}->{k} (basisTransformation)", (m + k) * B,
              lambda r, m=m, C=C: V.linearCombinationBlock(pick(r, m), C), reps=10)
    for m in (8, 16, 32):
        for method in ("cgs2", "mgs"):
            passes = 2 if method == "cgs2" else 1
            # cgs2: 2 x (multi_dot + multi_axpy) = 2 x ((m+1) + (m+2)) vector passes; mgs (round 4, one fused kernel per
            # vector): first kernel x + q (2), middle kernels x, q_prev, q_next + x written (4 each), last x, q_prev + x (3)
            nbytes = (2 * ((m + 1) + (m + 2)) if method == "cgs2" else 2 + 4 * (m - 1) + 3) * B
            x = ea.HipVector(host.copy(), {"orthogonalization": method})
            timed(f"orthogonalize_against_set m={m} {method}", nbytes,
                  lambda r, m=m, x=x: V.orthogonalize_against_set(x, pick(r, m)), reps=6)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rep, open(out, "w"), indent=1)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()

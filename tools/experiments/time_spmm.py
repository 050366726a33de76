import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, eigensolvers_amd as ea
ctx = ea.HipContext.default()
for N, R in ((1_000_000, 32), (10_000_000, 64)):
    H = ea.HipCsrOperator.generate(N, R, seed=7)
    X = [ea.HipVector(np.random.default_rng(j).standard_normal(N)) for j in range(8)]
    for k in (8,):
        bufs = [x._buf for x in X[:k]]
        H.apply_block(bufs); ctx.synchronize()
        ctx.timer_start()
        for _ in range(5): Y = H.apply_block(bufs)
        tb = ctx.timer_stop() / 5
        y = ctx.alloc(N); H.apply(bufs[0], y); ctx.synchronize()
        ctx.timer_start()
        for _ in range(5):
            for b in bufs: H.apply(b, y)
        ts = ctx.timer_stop() / 5
        print(f"N={N} k={k}: block product {tb:.3f} ms (pack+spmm+unpack), {k} single products {ts:.3f} ms, ratio {ts/tb:.2f}")
    del H, X

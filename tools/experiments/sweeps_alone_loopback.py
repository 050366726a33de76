#!/usr/bin/env python3
"""One rank's launches of a partitioned product ALONE on the GPU, for P loopback ranks in one process (the process guard of
this pool stops real processes at 6; 8 contexts in one process are fine): every rank builds its slab of the N = 1e7 operator,
all make a few real products, then the ranks take turns - exchange switched off (hipeig_comm_set_exchange), the others parked
at a host barrier.  usage: sweeps_alone_loopback.py P [N [nnz_row]]"""
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
from eigensolvers_amd.distributed import LoopbackGroup, row_range

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
grp = LoopbackGroup(P)
bar = threading.Barrier(P)


def body(rank, ctx):
    b, e = row_range(N, P, rank)
    H = ea.HipCsrOperator.generate(N, R, seed=7, row_begin=b, row_end=e, ctx=ctx)
    x = ea.HipVector(np.random.default_rng(100 + rank).standard_normal(e - b), ctx=ctx)
    y = ctx.alloc(e - b)
    for _ in range(3):
        H.apply_shifted(0.02, x._buf, y)                 # real products: layouts built, gathered buffer filled
    ctx.synchronize()
    ms = None
    for r in range(P):
        bar.wait()
        if r == rank:
            ctx.set_exchange(False)
            for _ in range(2):
                H.apply_shifted(0.02, x._buf, y)
            ctx.timer_start()
            for _ in range(10):
                H.apply_shifted(0.02, x._buf, y)
            ms = ctx.timer_stop() / 10
            ctx.set_exchange(True)
            ctx.synchronize()
    bar.wait()
    return ms, H.layout_info()


try:
    res = grp.run(body)
finally:
    grp.close()
lay = res[0][1]
print(f"P = {P}, N = {N}: sweeps alone per rank (ms) " + " ".join(f"{m:.4f}" for m, _ in res) +
      f" | max {max(m for m, _ in res):.4f}, ideal {1.90 / P:.3f} | column splits {lay['column_splits']}, "
      f"exchange chunks {lay['exchange_chunks']}, rows per block {lay['rows_per_block']}", flush=True)

#!/usr/bin/env python3
"""Sweep time of ONE rank's row slab of the N = 1e7 operator (no communication): the compute part of a
P-GPU strong-scaling step.  usage: slab_time.py P [N] [nnz_row]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eigensolvers_amd as ea  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ctx = ea.HipContext.default()
rows = N // P
H = ea.HipCsrOperator.generate(N, R, seed=7, row_begin=0, row_end=rows)
x = ctx.alloc(N)
y = ctx.alloc(rows)
ea._lib.call("hipeig_vec_fill", ctx.handle, x.ptr, N, 1.0)
for _ in range(3):
    H.apply_shifted(0.02, x, y)
ctx.synchronize()
ctx.timer_start()
reps = 20
for _ in range(reps):
    H.apply_shifted(0.02, x, y)
ms = ctx.timer_stop() / reps
print(f"P={P} rows={rows} nnz={H.nnz} variant={H.last_variant()} launches={H.launches_per_apply()} "
      f"CSPLIT={os.environ.get('HIPEIG_TCOOW_CSPLIT', 'auto')}: {ms:.4f} ms per slab product "
      f"(1/P of the single-GPU 2.13 ms = {2.13 / P:.3f} ms)", flush=True)

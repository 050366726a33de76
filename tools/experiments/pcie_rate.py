#!/usr/bin/env python3
"""Host <-> device rates at the boundary: HipVector(array) (upload) and .array (download), N = 1e7."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eigensolvers_amd as ea
N = 10_000_000
h = np.random.default_rng(0).standard_normal(N)
v = ea.HipVector(h)          # warm up
_ = v.array
t = time.perf_counter(); reps = 10
for _ in range(reps):
    v = ea.HipVector(h)
ea.HipContext.default().synchronize()
up = (time.perf_counter() - t) / reps
t = time.perf_counter()
for _ in range(reps):
    a = v.array
down = (time.perf_counter() - t) / reps
print(f"upload {8e-9 * N / up:.1f} GB/s ({up * 1e3:.2f} ms per 80 MB vector, pageable host memory), "
      f"download {8e-9 * N / down:.1f} GB/s ({down * 1e3:.2f} ms)")

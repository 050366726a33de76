#!/usr/bin/env python3
"""Block complex-shift product (hipeig_spmm_shift_pairs): 4 complex operands per pass (8-wide block, 64 B per operand row)
against 8 (16 wide, 128 B = one line per gather), per block kernel.  Checks every product against the single pair products.
usage: pair_block_width.py N nnz_row [npairs]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
N, R = int(sys.argv[1]), int(sys.argv[2])
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = ea.HipContext.default()
H = ea.HipCsrOperator.generate(N, R, seed=7)
rng = np.random.default_rng(3)
xs = []
for p in range(NP):
    xr, xi = ctx.alloc(N), ctx.alloc(N)
    for b in (xr, xi):
        host = np.ascontiguousarray(rng.standard_normal(N))          # kept alive across the call
        ea._lib.call("hipeig_vec_upload", ctx.handle, b.ptr, host.ctypes.data, N)
    xs.append((xr, xi))
z = 0.05 + 0.11j
ref = []
for xr, xi in xs:
    yr, yi = ctx.alloc(N), ctx.alloc(N)
    ea._lib.call("hipeig_spmv_shift_pair", ctx.handle, H.handle, z.real, z.imag, 1.0, xr.ptr, xi.ptr, yr.ptr, yi.ptr)
    ref.append((ea.HipVector(yr).array, ea.HipVector(yi).array))
ctx.synchronize()
ctx.timer_start()
for _ in range(3):
    for xr, xi in xs:
        yr, yi = ctx.alloc(N), ctx.alloc(N)
        ea._lib.call("hipeig_spmv_shift_pair", ctx.handle, H.handle, z.real, z.imag, 1.0, xr.ptr, xi.ptr, yr.ptr, yi.ptr)
print(f"N {N}: single pair products {ctx.timer_stop() / 3 / NP:.4f} ms per complex operand", flush=True)
for variant in (0, 1, 2):
    for width in (4, 8):
        os.environ["HIPEIG_PAIR_BLOCK_WIDTH"] = str(width)
        H.set_block_variant(variant)
        try:
            ys = H.apply_shifted_pairs(z, xs)
        except Exception as exc:
            print(f"variant {variant} width {width}: {exc}")
            continue
        err = max(max(np.max(np.abs(ea.HipVector(y[0]).array - r[0])), np.max(np.abs(ea.HipVector(y[1]).array - r[1])))
                  for y, r in zip(ys, ref))
        scale = max(np.max(np.abs(r[0])) for r in ref)
        ctx.synchronize()
        reps = 5
        ctx.timer_start()
        for _ in range(reps):
            ys = H.apply_shifted_pairs(z, xs)
        ms = ctx.timer_stop() / reps
        print(f"N {N} block kernel {H.block_info()['variant']:22s} (set {variant}) {width} complex operands per pass: "
              f"{ms / NP:.4f} ms per complex operand, max |diff| / max|y| = {err / scale:.2e}", flush=True)

"""Complex product y = z*x - H x (the GCROT matvec of FEAST's contour solves): one pair sweep against the two real
sweeps + two updates it replaces.  python tools/pair_bench.py [N [nnz_row]]  -> one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eigensolvers_amd as ea

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nnz_row = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if N > 2_000_000 else 32)
ctx = ea.HipContext.default()
H = ea.HipCsrOperator.generate(N, nnz_row, seed=7)
rng = np.random.default_rng(0)
xr, xi = ea.HipVector(rng.standard_normal(N)), ea.HipVector(rng.standard_normal(N))
yr, yi, y1 = ctx.alloc(N), ctx.alloc(N), ctx.alloc(N)
z = 0.02 + 0.11j


def timed(fn, reps):
    fn(); ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


reps = 20 if N > 2_000_000 else 100
out = {"N": N, "nnz": int(H.nnz), "z": [z.real, z.imag]}
out["single_product_ms"] = round(timed(lambda: H.apply_shifted(z.real, xr._buf, y1), reps), 4)
forced = os.environ.get("HIPEIG_PAIR_SWEEP")          # HIPEIG_PAIR_SWEEP=1 from outside: time the pair sweep where the rule would not take it
os.environ["HIPEIG_PAIR_SWEEP"] = "0"
out["two_sweeps_ms"] = round(timed(lambda: H.apply_shifted_pair(z, xr._buf, xi._buf, yr, yi), reps), 4)
a = ea.HipVector(yr).array.copy(), ea.HipVector(yi).array.copy()
if forced is None:
    del os.environ["HIPEIG_PAIR_SWEEP"]
else:
    os.environ["HIPEIG_PAIR_SWEEP"] = forced
out["pair_sweep_ms"] = round(timed(lambda: H.apply_shifted_pair(z, xr._buf, xi._buf, yr, yi), reps), 4)
out["pair_info"] = H.pair_info()
b = ea.HipVector(yr).array, ea.HipVector(yi).array
out["max_rel_diff"] = float(max(np.max(np.abs(a[0] - b[0])), np.max(np.abs(a[1] - b[1]))) / max(np.max(np.abs(a[0])), np.max(np.abs(a[1]))))
out["speedup"] = round(out["two_sweeps_ms"] / out["pair_sweep_ms"], 3)
# algorithmic bytes of the complex product: the stream once, both operand halves once, both results (SURVEY 8d form)
out["algorithmic_GB"] = round((H.nnz * 12 + (N + 1) * 4 + 2 * 8 * N + 2 * 8 * N) / 1e9, 3)
out["pair_TBps"] = round(out["algorithmic_GB"] / out["pair_sweep_ms"], 3)
print(json.dumps(out))

"""One MINRES solve at N = 2e5 (the size where the captured chunk helps most) - used by graph_rocprof.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
NNZ = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H = ea.HipCsrOperator.generate(N, NNZ, seed=7)
w0 = ea.HipVector(np.ones(N)).applyOp(H)          # loads the runtime before the map is printed
for line in open("/proc/self/maps"):
    if " r-xp " in line and any(k in line for k in ("libamdhip64", "librocprofiler", "libhsa-runtime", "libhipeig", "libc.so", "rocprofv3", "libroctx", "librocprofiler-sdk-tool")):
        print("MAP", line.split()[0], line.split()[-1], file=sys.stderr)
b = ea.HipVector(np.random.default_rng(1).standard_normal(N), {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}})
if os.environ.get("PRE_GRAM"):                                           # what the Lanczos driver does before its first solve
    b.normalize()
    S = ea.HipVector.overlapMatrix([b]); M = ea.HipVector.matrixRepresentation(H, [b])
    print("pre-gram", S, M, file=sys.stderr)
w = ea.HipVector.solve(H, b, 0.02)
for extra in range(int(os.environ.get("EXTRA_SOLVES", "0"))):          # graph re-use across solves, vector work in between
    b2 = w.copy().normalize()
    s = b2.vdot(w)
    w = ea.HipVector.solve(H, b2, 0.02)
    print("extra solve", extra, w.last_solve_stats["iterations"], file=sys.stderr)
print("graph", os.environ.get("HIPEIG_GRAPH", "0"), "iterations", w.last_solve_stats["iterations"], "istop", w.last_solve_stats["istop"], "|w|", w.norm())

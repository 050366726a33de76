import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import eigensolvers_amd as ea
from eigensolvers_amd.generators import gapped_csr_host
Hh = gapped_csr_host(4000, 32, seed=7).tocsr()
x = np.random.default_rng(31).standard_normal(4000)
bound = abs(Hh).sum(axis=1).max(); xmax = np.abs(x).max()
ex = math.frexp(bound * xmax)[1]; S = 2.0 ** (61 - ex)
print("bound", bound, "xmax", xmax, "S = 2^", 61 - ex)
emu = np.zeros(4000)
for i in range(4000):
    s, e = Hh.indptr[i], Hh.indptr[i + 1]
    q = np.rint(Hh.data[s:e] * x[Hh.indices[s:e]] * S)
    emu[i] = float(int(sum(int(v) for v in q))) / S
ys = []
for b in range(3):
    H = ea.HipCsrOperator.from_scipy(Hh); H.set_variant(5)
    ys.append(ea.HipVector(x).applyOp(H).array)
for b in range(3):
    bad = np.nonzero(ys[b] != emu)[0]
    print("build", b, "rows != emulation:", len(bad), bad[:8], [(ys[b][i] - emu[i]) * S for i in bad[:8]])
print("build0 vs build1 differ:", int(np.sum(ys[0] != ys[1])), " build1 vs build2:", int(np.sum(ys[1] != ys[2])))
H.set_variant(4); y4 = ea.HipVector(x).applyOp(H).array
print("variant 4 vs emulation: max abs", np.max(np.abs(y4 - emu)))
print("---- test flow: build, apply x2, apply_shifted, rebuild ...")
ctx = ea.HipContext.default()
outs = []
for b in range(3):
    H = ea.HipCsrOperator.from_scipy(Hh); H.set_variant(5)
    y1 = ea.HipVector(x).applyOp(H).array
    y2 = ea.HipVector(x).applyOp(H).array
    buf = ctx.alloc(4000)
    H.apply_shifted(0.02, ea.HipVector(x)._buf, buf)
    ysh = ea.HipVector(buf).array
    outs.append(y1)
    print("build", b, "y1==y2", np.array_equal(y1, y2), "y1 != emu:", int(np.sum(y1 != emu)), "shift err", np.max(np.abs(ysh - (0.02 * x - emu))))

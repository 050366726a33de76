import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import eigensolvers_amd as ea
from eigensolvers_amd.generators import gapped_csr_host
# --- generator diff
Hd = ea.HipCsrOperator.generate(4000, 32, seed=7).to_scipy(); Hh = gapped_csr_host(4000, 32, seed=7)
d = np.nonzero(Hd.data != Hh.data)[0]
print("gen: differing", len(d), "of", Hh.nnz)
rows = np.searchsorted(Hh.indptr, d, side="right") - 1
for p, r in list(zip(d, rows))[:10]:
    print("  p", p, "row", r, "col", Hh.indices[p], "dev", repr(Hd.data[p]), "host", repr(Hh.data[p]), "diag" if Hh.indices[p] == r else "")
print("  all diffs on diagonal:", bool(np.all(Hh.indices[d] == rows)))
# --- ragged spmv
rng = np.random.default_rng(5); n = 3000; rows_ = []
for i in range(n):
    k = 0 if i % 7 == 0 else 2600 if i == 11 else n if i == 1500 else int(rng.integers(1, 90))
    rows_.append((rng.integers(0, n, size=k), rng.standard_normal(k)))
rowptr = np.concatenate([[0], np.cumsum([len(c) for c, _ in rows_])]).astype(np.int64)
col = np.concatenate([c for c, _ in rows_]).astype(np.int32); val = np.concatenate([v for _, v in rows_])
A = sp.csr_matrix((val, col, rowptr), shape=(n, n)); x = rng.standard_normal(n); ref = A @ x
H = ea.HipCsrOperator.from_csr_arrays(rowptr, col, val, n)
import ctypes as C
info = (C.c_int64 * 8)(); ea._lib.call("hipeig_csr_info", H.handle, info); print("csr info", list(info))
for variant in (1, 2):
    H.set_variant(variant)
    got = ea.HipVector(x).applyOp(H).array
    bad = np.nonzero(np.abs(got - ref) > 1e-10)[0]
    print("variant", variant, "bad rows", len(bad), bad[:20], "first vals", got[bad[:3]], ref[bad[:3]])
# without the long rows
for drop in ([11], [1500], [11, 1500]):
    keep = np.ones(n, bool); keep[drop] = False
    A2 = A.copy().tolil()
    for r in drop: A2.rows[r] = []; A2.data[r] = []
    A2 = A2.tocsr(); H2 = ea.HipCsrOperator.from_scipy(A2); ref2 = A2 @ x
    for variant in (1, 2):
        H2.set_variant(variant); got = ea.HipVector(x).applyOp(H2).array
        print("drop", drop, "variant", variant, "bad", int(np.sum(np.abs(got - ref2) > 1e-10)))

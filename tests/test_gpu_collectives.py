"""The RCCL code path on ONE GPU: a one-rank communicator with HIPEIG_FORCE_COLLECTIVES=1
routes every reduction through ncclAllReduce and every operator application through the
in-place ncclAllGather + column remap, exactly as an N-rank run does.  (Multi-rank runs need
several GPUs; the driver's scaling bench covers those.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO, load_golden

pytestmark = pytest.mark.gpu

CHILD = r"""
import json, sys, numpy as np
sys.path.insert(0, %(repo)r)
import eigensolvers_amd as ea
from eigensolvers_amd.generators import gapped_csr_host, guess_vector
ctx = ea.HipContext(0); ea.HipContext._default = ctx
ctx.attach_comm(1, 0, ctx.new_unique_id())
N = 4000
H = ea.HipCsrOperator.generate(N, 32, seed=7, ctx=ctx)
Hh = gapped_csr_host(N, 32, seed=7)
x = np.random.default_rng(3).standard_normal(N)
out = {}
for variant in (1, 2, 3, 4):
    H.set_variant(variant)
    y = ea.HipVector(x, ctx=ctx).applyOp(H).array
    out["spmv_err_%%d" %% variant] = float(np.max(np.abs(y - Hh @ x)))
H.set_variant(0)
X = ea.HipVector(x, ctx=ctx)
out["dot_err"] = abs(X.vdot(X) - float(np.dot(x, x)))
opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}}
ev, Y, st = ea.inexactLanczosDiagonalization(H, ea.HipVector(guess_vector(N, 1).copy(), opts, ctx=ctx),
                                             0.02, 8, 10, 1e-13, writeOut=False)
out.update(ev0=float(ev[0]), cumIter=int(st["cumIter"]), conv=bool(st["isConverged"]))
# a larger operator (3 column windows): the all-gather overlaps the local-window launch, the remaining
# window is swept afterwards from the gathered operand; compare with the unsplit CSR-stream kernel
N2 = 300000
H2 = ea.HipCsrOperator.generate(N2, 32, seed=5, ctx=ctx)
x2 = ea.HipVector(np.random.default_rng(4).standard_normal(N2), ctx=ctx)
H2.set_variant(2); y_ref = x2.applyOp(H2)
H2.set_variant(4); y_ov = x2.applyOp(H2)
d = ea.HipVector.linearCombination([y_ov, y_ref], [1.0, -1.0])
out["overlap_spmv_rel"] = d.norm() / y_ref.norm()
b2 = ea.HipVector(guess_vector(N2, 2).copy(), opts, ctx=ctx); b2.normalize()
w4 = ea.HipVector.solve(H2, b2, 0.02); it4 = w4.last_solve_stats["iterations"]
H2.set_variant(2)
w2 = ea.HipVector.solve(H2, b2, 0.02); it2 = w2.last_solve_stats["iterations"]
d = ea.HipVector.linearCombination([w4, w2], [1.0, -1.0])
out.update(overlap_minres_rel=d.norm() / w2.norm(), it4=it4, it2=it2, coll=w2.last_solve_stats["collectives"])
# the backend choice bench.py makes per operator on a multi-GPU node (RCCL all-gather vs direct peer writes), here with
# one rank: both backends run the same changing operands, results compared product by product, the faster one kept
from eigensolvers_amd import distributed as D
assert D.enable_direct_gather(ctx, D.gathered_capacity(N2, 1), 0, 1)
H3 = ea.HipCsrOperator.generate(N2, 32, seed=5, ctx=ctx)          # created with the direct buffers in place
out["choice"] = D.choose_gather_backend(ctx, H3, D.DeviceGroup(ctx), reps=3)
out["gather_info"] = ctx.gather_info()
ctx.set_gather_backend("rccl"); ctx.set_allreduce_backend("rccl")
# ... and the same choice when the direct backend FAILS its trial (it has never run between different devices): an
# injected error inside the trial - every rank falls back to RCCL together, the buffers are released, products go on
if not ctx.gather_info()["direct_attached"]:
    assert D.enable_direct_gather(ctx, D.gathered_capacity(N2, 1), 0, 1)
orig = ctx.set_gather_backend
def failing(name):
    if name == "direct":
        raise RuntimeError("injected failure of the direct trial")
    return orig(name)
ctx.set_gather_backend = failing
out["failed_choice"] = D.choose_gather_backend(ctx, H3, D.DeviceGroup(ctx), reps=2)
ctx.set_gather_backend = orig
out["gather_info_after_failure"] = ctx.gather_info()
y_after = x2.applyOp(H3)
d = ea.HipVector.linearCombination([y_after, y_ref], [1.0, -1.0])
out["product_after_failure_rel"] = d.norm() / y_ref.norm()
import ctypes
buf = ctypes.create_string_buffer(512)
ea._lib.call("hipeig_comm_library", buf, 512)
out["rccl"] = buf.value.decode()
out["torch_loaded"] = "torch" in sys.modules
print("RESULT " + json.dumps(out))
"""


def test_forced_collectives_single_rank():
    env = dict(os.environ, HIPEIG_FORCE_COLLECTIVES="1")
    p = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    r = json.loads(line[7:])
    for variant in (1, 2, 3, 4):
        assert r["spmv_err_%d" % variant] < 1e-12
    assert r["dot_err"] < 1e-9
    g = load_golden("gapped_csr_n4000_minres.npz")
    assert abs(r["ev0"] - g["ev"][0]) <= 1e-10 * abs(g["ev"][0])
    assert r["cumIter"] == int(g["cumIter"]) and r["conv"]
    assert r["overlap_spmv_rel"] < 1e-14
    assert r["it4"] == r["it2"] and r["overlap_minres_rel"] < 1e-8
    enq = 16 * -(-(r["it2"] + 1) // 16)                      # the stop of iteration K is seen in KC of iteration K + 1
    assert r["coll"] == 2 * enq                              # one operand exchange (carrying <y,y>) + one all-reduce per iteration
    ch = r["choice"]
    assert ch["results_agree"] and ch["products_compared"] == 6 and ch["chosen"] in ("rccl", "direct")
    assert ch["rccl_ms"] > 0 and ch["direct_ms"] > 0 and ch["allreduce_rccl_ms"] > 0 and ch["allreduce_direct_ms"] > 0
    assert ch["allreduce_results_agree"] and r["gather_info"]["wait_error"] == 0
    # the two gathered buffers per rank stay only if one of the two direct backends won
    assert r["gather_info"]["direct_attached"] == ("direct" in (ch["chosen"], ch["allreduce_chosen"]))
    fc = r["failed_choice"]
    assert fc["chosen"] == "rccl" and fc["allreduce_chosen"] == "rccl" and "injected failure" in fc["direct_trial_failed"]
    assert not r["gather_info_after_failure"]["direct_attached"] and r["gather_info_after_failure"]["backend"] == "rccl"
    assert r["product_after_failure_rel"] < 1e-14
    # the collectives run on ROCm's own RCCL, whatever else the process has loaded; no torch on the product path
    assert r["rccl"].startswith("/opt/rocm") and not r["torch_loaded"], r["rccl"]

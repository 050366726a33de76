"""The direct peer-write exchange (csrc/comm_direct.hip) between REAL processes: three ranks started by the package's own
launcher share the one GPU of the test box, map each other's buffers through hipIpc handles that travel over the TCP
group, and run row-partitioned products, dots, MINRES and block calls with no RCCL at all (`hipeig_comm_init_direct`).  What a
multi-GPU node adds to this is only the xGMI hop under the same stores and flags."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
import eigensolvers_amd as ea
from eigensolvers_amd import distributed as D
from eigensolvers_amd.generators import guess_vector

rank, world, _ = D.world_from_env()
N = %(N)d
ctx = ea.HipContext(0)                                   # every rank on the box's one GPU
whole_ctx = ea.HipContext(0)                             # a second context without a communicator: the reference
D.attach_direct(ctx, D.gathered_capacity(N, world), rank, world)
assert "torch" not in sys.modules
info0 = ctx.gather_info()
b, e = D.row_range(N, world, rank)
H = ea.HipCsrOperator.generate(N, 24, seed=21, row_begin=b, row_end=e, ctx=ctx)
whole = ea.HipCsrOperator.generate(N, 24, seed=21, ctx=whole_ctx)
whole.set_variant(2)
x = np.random.default_rng(5).standard_normal(N)
y_ref = ea.HipVector(x, ctx=whole_ctx).applyOp(whole).array
out = {"rank": rank, "world": world, "backend": info0["backend"], "allreduce_backend": info0["allreduce_backend"]}
X = ea.HipVector(x[b:e], ctx=ctx)
for variant in (2, 4):
    H.set_variant(variant)
    errs = []
    for rep in range(6):                                 # both buffers of the double-buffered exchange, several times
        y = X.applyOp(H).array
        errs.append(float(np.max(np.abs(y - y_ref[b:e])) / np.max(np.abs(y_ref))))
    out["spmv%%d" %% variant] = max(errs)
out["layout"] = H.layout_info()
out["dot"] = X.vdot(X)
out["dot_ref"] = float(np.dot(x, x))
opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}}
bf = guess_vector(N, 4) / np.linalg.norm(guess_vector(N, 4))
w_ref = ea.HipVector.solve(whole, ea.HipVector(bf.copy(), dict(opts), ctx=whole_ctx), 0.02)
w = ea.HipVector.solve(H, ea.HipVector(bf[b:e].copy(), dict(opts), ctx=ctx), 0.02)
out.update(it=w.last_solve_stats["iterations"], it_ref=w_ref.last_solve_stats["iterations"],
           istop=w.last_solve_stats["istop"], coll=w.last_solve_stats["collectives"],
           werr=float(np.linalg.norm(w.array - w_ref.array[b:e]) / np.linalg.norm(w_ref.array)))
# block operands have no direct exchange: the block product runs as k single products, solveBlock one by one
Yb = H.apply_block([X._buf, X._buf])
out["block"] = max(float(np.max(np.abs(ea.HipVector(yb).array - y_ref[b:e])) / np.max(np.abs(y_ref))) for yb in Yb)
Wb = ea.HipVector.solveBlock(H, [ea.HipVector(bf[b:e].copy(), dict(opts), ctx=ctx) for _ in range(3)], 0.02)
out["block_it"] = [wb.last_solve_stats["iterations"] for wb in Wb]
g = D.DeviceGroup(ctx)
g.barrier()
out["allmax"] = g.allmax(float(rank))
info = ctx.gather_info()
out.update(exchanges=info["exchanges"], wait_error=info["wait_error"])
with open(os.path.join(%(outdir)r, "rank%%d.json" %% rank), "w") as f:
    json.dump(out, f)
D.tcp_group().barrier()                                  # nobody unmaps a buffer a peer may still write to
'''


@pytest.mark.timeout(900)
@pytest.mark.parametrize("chunks", [1, 2])
def test_three_processes_exchange_by_peer_writes(tmp_path, chunks):
    from eigensolvers_amd.distributed import launch_local
    N, P = 250_001, 3
    prog = tmp_path / "child.py"
    prog.write_text(CHILD % {"repo": REPO, "N": N, "outdir": str(tmp_path)})
    rc, out = launch_local([str(prog)], P, timeout=600,
                           env_extra={"HIPEIG_GATHER_CHUNKS": str(chunks), "HIPEIG_TCOOW_WBITS": "13", "HIPEIG_DIRECT_WAIT_S": "20"})
    assert rc == 0, out[-2000:]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(P)]
    for r, o in enumerate(res):
        assert (o["rank"], o["world"]) == (r, P) and o["backend"] == "direct" and o["allreduce_backend"] == "direct"
        assert o["layout"]["exchange_chunks"] == chunks
        assert o["spmv2"] < 1e-14 and o["spmv4"] < 1e-14
        assert abs(o["dot"] - o["dot_ref"]) < 1e-10 * o["dot_ref"] and o["dot"] == res[0]["dot"]     # identical on every rank
        assert o["it"] == res[0]["it"] and abs(o["it"] - o["it_ref"]) <= 2 and o["istop"] in (1, 2)
        assert o["coll"] == 2 * 16 * -(-(o["it"] + 1) // 16)
        assert o["werr"] < 1e-8
        assert o["block"] < 1e-14 and o["block_it"] == [o["it"]] * 3
        assert o["allmax"] == float(P - 1) and o["wait_error"] == 0 and o["exchanges"] > 12


FEAST_CHILD = r'''
import json, os, sys, warnings
import numpy as np
sys.path.insert(0, %(repo)r)
import eigensolvers_amd as ea
from eigensolvers_amd import distributed as D

rank, world, _ = D.world_from_env()
g = np.load(%(golden)r)
ctx = ea.HipContext(0)
D.attach_direct(ctx, D.gathered_capacity(g["A"].shape[0], world), rank, world)
comm = D.ContourReplicas(ctx)                            # replica mode first, then operator and vectors
opts = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-2}}
A = ea.HipCsrOperator.from_dense(g["A"], ctx=ctx)
Y = [ea.HipVector(g["guess"][:, i].copy(), dict(opts), ctx=ctx) for i in range(6)]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ev, Yf, st = ea.feastDiagonalization(A, Y, 8, "legendre", 160.0, 166.0, 1e-10, 20, writeOut=False, contourComm=comm)
info = ctx.gather_info()
with open(os.path.join(%(outdir)r, "rank%%d.json" %% rank), "w") as f:
    json.dump({"ev": [float(e) for e in ev], "it": int(st["outerIter"]), "n": len(Yf), "y0": Yf[0].array.tolist(),
               "wait_error": info["wait_error"], "allreduce_backend": info["allreduce_backend"]}, f)
D.tcp_group().barrier()
'''


@pytest.mark.timeout(900)
def test_feast_contour_points_on_three_processes(tmp_path):
    """BASELINE config #5's mapping - contour points dealt to the ranks, whole operator on each, one all-reduce per filtered
    vector - on three REAL processes joined by the RCCL-free communicator: the reference's own FEAST case
    (tests/golden/feast_n100.npz, 4 half-contour points -> 2 + 1 + 1), same iteration count and window eigenvalues as the
    reference's run, identical data on every rank."""
    from conftest import GOLDEN
    from eigensolvers_amd.distributed import launch_local
    P = 3
    golden = os.path.join(GOLDEN, "feast_n100.npz")
    prog = tmp_path / "feast_child.py"
    prog.write_text(FEAST_CHILD % {"repo": REPO, "golden": golden, "outdir": str(tmp_path)})
    rc, out = launch_local([str(prog)], P, timeout=600, env_extra={"HIPEIG_DIRECT_WAIT_S": "20"})
    assert rc == 0, out[-2000:]
    g = np.load(golden)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(P)]
    inside = (g["ev"] >= 160.0) & (g["ev"] <= 166.0)
    assert inside.sum() == 3
    for o in res:
        assert o["wait_error"] == 0 and o["allreduce_backend"] == "direct"
        assert o["ev"] == res[0]["ev"] and o["y0"] == res[0]["y0"]                 # every replica ends with the same data
        assert o["it"] == int(g["outerIter"]) and o["n"] == int(g["nvec"])
        np.testing.assert_allclose(np.asarray(o["ev"])[inside], g["ev"][inside], rtol=1e-9)


@pytest.mark.timeout(900)
def test_bench_started_plainly_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` exactly as a driver would start it - no launcher, no environment - with the RCCL-free
    communicator (RCCL refuses two ranks on one device): the parent spawns the ranks, they exchange by peer writes, and
    rank 0's ONE JSON line carries what the first real scaling run needs to explain itself: ranks seen through the
    communicator, the exchange backend, per-phase times, and a Lanczos run that converges to the single-GPU value."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HIPEIG_COMM"] = "direct"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--n", "2000000", "--steps", "5",
                        "--warmup", "2", "--no-cpu"], capture_output=True, text=True, env=env, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["scaling"] == "strong"
    assert out["config"]["exchange"]["chosen"] == "direct" and out["config"]["rccl_library"] is None
    ph = out["phases"]
    assert ph["gather_ms"] > 0 and ph["product_ms"] > 0 and ph["allreduce_ms"] > 0 and ph["exchange_chunks"] >= 1
    assert ph["local_sweep_ms"] is not None and ph["remote_sweep_ms"] is not None          # the overlap path ran
    lz = out["lanczos"]
    assert lz["converged"] and lz["true_residual_norm"] < 1e-6 and abs(lz["ritz_value"] - 0.2 / 15) < 2e-3
    assert out["value"] > 0 and out["roofline"]["frac"] > 0

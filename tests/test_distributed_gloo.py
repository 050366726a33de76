"""world_size-2 CPU rehearsal (gloo, test-only stand-in for RCCL) of the row-partitioned path: the product's host driver
runs on every rank with local slices, all ranks must take identical decisions and the
result must equal the single-process oracle run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_blas_thread():
    """Several ranks on the container's 8 cores: one BLAS thread each (n = 100 .. 1500 gains nothing from more, and the
    oversubscribed pools cost tens of seconds)."""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass


def _worker(rank, world, port, N, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import eigensolvers_amd as ea
    from eigensolvers_amd import distributed as D
    from eigensolvers_amd.generators import gapped_csr_host, guess_vector
    from _dist_vector import DistRefVector, SlabOperator
    import torch.distributed as dist                       # test infrastructure only: the CPU stand-in collectives
    dist.init_process_group(backend="gloo")
    assert D.world_from_env() == (rank, world, rank)
    # the product's own rendezvous (stdlib TCP, no torch): how RCCL's 128-byte unique id travels
    assert D.exchange_bytes(bytes(range(128)) if rank == 0 else b"", 128, rank, world) == bytes(range(128))
    ea.AbstractVector.register(DistRefVector)
    ranges = D.all_row_ranges(N, world)
    b, e = ranges[rank]
    H = gapped_csr_host(N, 16, seed=3)                      # every rank can build its slab alone:
    slab = gapped_csr_host(N, 16, seed=3, row_begin=b, row_end=e)
    assert abs(slab - H[b:e]).max() == 0.0
    op = SlabOperator(H, ranges, rank)
    opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 3000, "linear_tol": 1e-10}}
    v0 = DistRefVector(guess_vector(N, 1, b, e).copy(), opts)
    ev, Y, st = ea.inexactLanczosDiagonalization(op, v0, 0.02, 6, 6, 1e-12, writeOut=False)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ev=ev, cumIter=st["cumIter"], conv=st["isConverged"],
             y0=Y[0].array, b=b, e=e)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_row_partition_matches_single_process(tmp_path):
    N, world = 1501, 2                                       # odd N -> unequal slabs (751 + 750)
    mp.spawn(_worker, args=(world, _free_port(), N, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    np.testing.assert_array_equal(r[0]["ev"], r[1]["ev"])    # identical control flow on every rank
    assert int(r[0]["cumIter"]) == int(r[1]["cumIter"]) and bool(r[0]["conv"])

    from oracle import lanczos_ref
    from oracle.numpy_vector import RefVector
    from eigensolvers_amd.generators import gapped_csr_host, guess_vector
    H = gapped_csr_host(N, 16, seed=3)
    opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 3000, "linear_tol": 1e-10}}
    ev, Y, st = lanczos_ref.inexact_lanczos(H, RefVector(guess_vector(N, 1).copy(), opts), 0.02, 6, 6, 1e-12)
    assert abs(r[0]["ev"][0] - ev[0]) <= 1e-10 * abs(ev[0])
    assert int(r[0]["cumIter"]) == st["cumIter"]
    full = np.concatenate([r[0]["y0"], r[1]["y0"]])
    assert abs(abs(np.dot(full, Y[0].array)) - 1) < 1e-8


# ---- FEAST contour replicas: contour points dealt to the ranks ---------------
class _GlooContour:
    """contourComm over gloo for ndarray vectors (test only)."""

    def __init__(self):
        import torch.distributed as dist
        self.rank, self.nranks = dist.get_rank(), dist.get_world_size()

    def allreduce(self, vec):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(vec.array, dtype=np.float64).copy())
        dist.all_reduce(t)
        return type(vec)(t.numpy(), vec.options)


def _feast_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _one_blas_thread()
    import eigensolvers_amd as ea
    from conftest import load_golden
    from eigensolvers_amd import distributed as D
    from oracle.numpy_vector import RefVector
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
    ea.AbstractVector.register(RefVector)
    g = load_golden("feast_n100.npz")
    opts = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-2}}
    Y = [RefVector(g["guess"][:, i].copy(), opts) for i in range(6)]
    ev, Yf, st = ea.feastDiagonalization(g["A"], Y, 8, "legendre", 160.0, 166.0, 1e-10, 20, writeOut=False,
                                         contourComm=_GlooContour())
    np.savez(os.path.join(out_dir, f"feast{rank}.npz"), ev=ev, outerIter=st["outerIter"], nvec=len(Yf),
             y0=Yf[0].array)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2])
def test_feast_contour_replicas_match_the_serial_run(tmp_path, world):
    """4 contour points over 2 ranks (2 + 2): every rank ends with the same eigenvalues, and they are
    the serial run's (the reference's own run, golden file).  Uneven splits (3 and 5 ranks) run on the
    GPU through the loopback group, tests/test_gpu_loopback.py."""
    mp.spawn(_feast_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from conftest import load_golden
    g = load_golden("feast_n100.npz")
    r = [np.load(tmp_path / f"feast{k}.npz") for k in range(world)]
    for k in range(world):
        np.testing.assert_array_equal(r[k]["ev"], r[0]["ev"])         # identical data after the all-reduce
        np.testing.assert_array_equal(r[k]["y0"], r[0]["y0"])
        np.testing.assert_allclose(r[k]["ev"], g["ev"], rtol=1e-9)
        assert int(r[k]["outerIter"]) == int(g["outerIter"]) and int(r[k]["nvec"]) == int(g["nvec"])


@pytest.mark.timeout(300)
def test_rendezvous_under_the_real_launcher(tmp_path):
    """bench.py --gpus N is started by ``python -m torch.distributed.run``; the product reads the launcher's
    environment and exchanges RCCL's 128-byte id over its own TCP channel without importing torch
    (tools/rendezvous_check.py asserts that in every rank).  Three ranks, the driver's command line."""
    import subprocess
    world = 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "tools", "rendezvous_check.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    for k in range(world):
        assert (tmp_path / f"rank{k}of{world}.txt").read_text().split() == [str(k), str(world), str(k), "1"]

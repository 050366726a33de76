"""Row partitioning and communicator bootstrap for the multi-GPU path (one process per GPU).

The operator is split into contiguous row slabs, one per rank;
every vector is split the same way.  The only exchanges on the path are an all-gather of
the operand slice before each operator application and a SUM all-reduce after each
reduction, both issued by ``libhipeig.so`` through RCCL on its compute stream.  The host
side only has to (a) agree on the row ranges and (b) distribute RCCL's 128-byte unique id,
which is done here with a stdlib TCP exchange (``exchange_bytes``); host-side barriers and the
few scalar reductions of a benchmark go through the library's own all-reduce
(``DeviceGroup``).  Nothing in this package imports torch.
"""
import contextlib
import os
import sys

# Multi-process GPU work on this driver stack needs dmabuf IPC (RCCL's intra-node transport fails with
# "hipIpcGetMemHandle: invalid argument" otherwise).  HSA reads the variable when the first HIP call
# initialises the runtime, so it has to be in place before a HipContext exists; an explicit setting wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


@contextlib.contextmanager
def stdout_to_stderr():
    """Send everything written to file descriptor 1 to stderr for the duration (gloo and RCCL
    print banners to stdout from native code; a benchmark's stdout must stay machine-readable)."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def row_range(N, nranks, rank):
    """Contiguous balanced slab [begin, end) of rank ``rank``; the first N % nranks ranks
    hold one extra row."""
    base, extra = divmod(int(N), int(nranks))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def all_row_ranges(N, nranks):
    return [row_range(N, nranks, r) for r in range(nranks)]


def world_from_env():
    """(rank, world_size, local_rank) from the torchrun environment (defaults: single)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


# ---- rendezvous: a stdlib TCP exchange of RCCL's 128-byte unique id ------------------------------
# Under ``torch.distributed.run`` the launcher's own store already listens on MASTER_PORT, so the
# exchange uses a port derived from it: rank 0 binds the first free one of a short deterministic
# candidate list, the other ranks try the candidates until one answers with the expected greeting.
_MAGIC = b"HIPEIG-RDZV-1"


def _candidate_ports():
    base = int(os.environ.get("HIPEIG_RDZV_PORT", "0"))
    if base:
        return [base]
    mp = int(os.environ.get("MASTER_PORT", "29511"))
    return [20000 + (mp * 7 + 13 + 101 * k) % 20000 for k in range(8)]


_exchange_seq = 0


def _run_token():
    return (os.environ.get("TORCHELASTIC_RUN_ID", "none") + ":" + os.environ.get("MASTER_PORT", "29511")).encode()


def exchange_bytes(payload, nbytes, rank, world, timeout=300.0):
    """Rank 0 hands ``payload`` (``nbytes`` bytes) to every other rank over TCP on MASTER_ADDR; returns it.
    No third-party module involved (the north star's "no PyTorch" covers the rendezvous too)."""
    import socket
    import time
    if world == 1:
        return bytes(payload)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    # the n-th exchange of this job only talks to the n-th exchange of the other ranks: a rank that is already one
    # exchange ahead is turned away (and retries) instead of being handed the previous record
    global _exchange_seq
    _exchange_seq += 1
    token = _MAGIC + b"|" + _run_token() + b"#" + str(_exchange_seq).encode()
    if rank == 0:
        srv = None
        for port in _candidate_ports():
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
                break
            except OSError:
                srv.close()
                srv = None
        if srv is None:
            raise RuntimeError(f"rendezvous: none of the ports {_candidate_ports()} is free on {addr}")
        srv.listen(world)
        srv.settimeout(timeout)
        served = set()
        try:
            while len(served) < world - 1:
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = conn.recv(256)
                    except OSError:
                        continue
                    if not hello.startswith(token + b"|"):
                        continue                              # a stranger on our port: ignore
                    peer = int(hello[len(token) + 1:].decode())
                    conn.sendall(token + b"|" + bytes(payload))
                    served.add(peer)
        finally:
            srv.close()
        return bytes(payload)
    deadline = time.time() + timeout
    while time.time() < deadline:
        for port in _candidate_ports():
            try:
                with socket.create_connection((addr, port), timeout=2.0) as c:
                    c.sendall(token + b"|" + str(rank).encode())
                    buf = b""
                    want = len(token) + 1 + nbytes
                    while len(buf) < want:
                        chunk = c.recv(want - len(buf))
                        if not chunk:
                            break
                        buf += chunk
                    if len(buf) == want and buf.startswith(token + b"|"):
                        return buf[len(token) + 1:]
            except OSError:
                pass
        time.sleep(0.05)
    raise TimeoutError(f"rendezvous: rank {rank} got no unique id from rank 0 within {timeout:.0f} s")


def attach_rccl(ctx, rank=None, world=None):
    """Create the RCCL communicator of ``ctx`` for the ranks of the launcher's environment (collective).
    Returns (rank, world)."""
    if rank is None or world is None:
        rank, world, _ = world_from_env()
    with stdout_to_stderr():
        uid = ctx.new_unique_id() if rank == 0 else b""
        uid = exchange_bytes(uid, 128, rank, world)
        ctx.attach_comm(world, rank, uid)
    return rank, world


class DeviceGroup:
    """Host-side helpers of a row-partitioned run built on the library's own collectives (RCCL through
    ``hipeig_vec_allreduce``): a barrier and small MAX / SUM reductions of host scalars.  ``ctx`` must have
    a communicator attached; with one rank everything is local."""

    def __init__(self, ctx):
        self.ctx, self.rank, self.world = ctx, ctx.rank, ctx.nranks
        self._buf = ctx.alloc(max(self.world, 1)) if self.world > 1 else None

    def allgather_scalar(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank (one SUM all-reduce of ``world`` doubles)."""
        import ctypes as C
        import numpy as np
        from . import _lib
        if self.world == 1:
            return [float(x)]
        host = np.zeros(self.world)
        host[self.rank] = float(x)
        _lib.call("hipeig_vec_upload", self.ctx.handle, self._buf.ptr, host.ctypes.data_as(C.c_void_p), self.world)
        _lib.call("hipeig_vec_allreduce", self.ctx.handle, self._buf.ptr, self.world)
        _lib.call("hipeig_vec_download", self.ctx.handle, host.ctypes.data_as(C.c_void_p), self._buf.ptr, self.world)
        return [float(v) for v in host]

    def allmax(self, x):
        return max(self.allgather_scalar(x))

    def allsum(self, x):
        return sum(self.allgather_scalar(x))

    def barrier(self):
        self.ctx.synchronize()
        self.allgather_scalar(0.0)


class ContourReplicas:
    """``contourComm`` of ``feastDiagonalization`` for HipVector: one contour point per GPU, whole
    operator and vectors on every rank, one RCCL all-reduce per filtered vector (SURVEY.md section
    8e).  Puts the context into replica mode: create the operator and the vectors AFTER this."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.rank, self.nranks = ctx.rank, ctx.nranks
        ctx.set_partitioned(False)

    def allreduce(self, vec):
        self.ctx.allreduce_vector(vec._buf)
        return vec


class LoopbackGroup:
    """``nranks`` contexts on ONE device joined by the library's in-process loopback collectives, each
    driven by its own host thread - a rehearsal of the row-partitioned path on a single GPU (RCCL
    refuses two ranks on one device).  ``run(fn)`` calls ``fn(rank, ctx)`` on every rank concurrently
    and returns the results in rank order; an exception on any rank is re-raised."""

    def __init__(self, nranks, device=0):
        import ctypes as C
        from . import _lib
        from .hip_vector import HipContext
        self.nranks = int(nranks)
        h = C.c_void_p()
        _lib.call("hipeig_loopback_group_create", self.nranks, C.byref(h))
        self.handle = h
        self.contexts = [HipContext(device) for _ in range(self.nranks)]
        for r, ctx in enumerate(self.contexts):
            ctx.attach_loopback(self.handle, self.nranks, r)

    def run(self, fn):
        import threading
        results, errors = [None] * self.nranks, [None] * self.nranks

        def work(r):
            try:
                results[r] = fn(r, self.contexts[r])
            except BaseException as exc:      # noqa: BLE001 - re-raised below
                errors[r] = exc

        threads = [threading.Thread(target=work, args=(r,), name=f"loopback-rank{r}") for r in range(self.nranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errors:
            if e is not None:
                raise e
        return results

    def close(self):
        from . import _lib
        if self.handle is not None:
            self.contexts = []
            _lib.call("hipeig_loopback_group_destroy", self.handle)
            self.handle = None

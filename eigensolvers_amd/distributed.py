"""Row partitioning and communicator bootstrap for the multi-GPU path (one process per GPU).

The operator is split into contiguous row slabs, one per rank;
every vector is split the same way.  The only exchanges on the path are an all-gather of
the operand slice before each operator application and a SUM all-reduce after each
reduction, both issued by ``libhipeig.so`` through RCCL on its compute stream.  The host
side only has to (a) agree on the row ranges and (b) distribute RCCL's 128-byte unique id,
which is done here over ``torch.distributed`` (gloo) - torch is plumbing for the
rendezvous, no tensor of the hot path ever goes through it.
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


def init_process_group_gloo():
    """Join the gloo group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo")
    return dist


def broadcast_bytes(payload, nbytes, src=0):
    """Broadcast a fixed-size byte string from ``src`` over the gloo group."""
    import torch
    import torch.distributed as dist
    buf = torch.zeros(nbytes, dtype=torch.uint8)
    if dist.get_rank() == src:
        buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8).clone()
    dist.broadcast(buf, src=src)
    return bytes(buf.numpy().tobytes())


def attach_rccl(ctx):
    """Create the RCCL communicator of ``ctx`` for the current gloo group (collective)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    with stdout_to_stderr():
        uid = ctx.new_unique_id() if rank == 0 else b"\0" * 128
        uid = broadcast_bytes(uid, 128, src=0)
        ctx.attach_comm(world, rank, uid)
    return rank, world


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

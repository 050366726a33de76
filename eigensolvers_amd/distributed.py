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


# ---- rendezvous: a stdlib TCP group ------------------------------------------------------------------
# Rank 0 listens, every other rank connects once and the sockets stay open for the life of the process; all that
# ever travels are small records (RCCL's 128-byte unique id, hipIpc memory handles of the direct all-gather, a few
# scalars of a benchmark).  Under ``torch.distributed.run`` the launcher's own store already listens on MASTER_PORT,
# so the group uses a port derived from it: rank 0 binds the first free one of a short deterministic candidate
# list, the other ranks try the candidates until one answers the greeting.  The listener binds MASTER_ADDR only
# (127.0.0.1 on one node); the greeting carries the launcher's run id and port, which keeps two jobs on one host
# apart - it is not an authentication scheme.
_MAGIC = b"HIPEIG-RDZV-2"


def _candidate_ports():
    base = int(os.environ.get("HIPEIG_RDZV_PORT", "0"))
    if base:
        return [base]
    mp = int(os.environ.get("MASTER_PORT", "29511"))
    return [20000 + (mp * 7 + 13 + 101 * k) % 20000 for k in range(8)]


def _run_token():
    return (os.environ.get("TORCHELASTIC_RUN_ID", "none") + ":" + os.environ.get("MASTER_PORT", "29511")).encode()


def _send_msg(sock, payload):
    sock.sendall(len(payload).to_bytes(8, "little") + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock, limit=1 << 26):
    n = int.from_bytes(_recv_exact(sock, 8), "little")
    if n > limit:
        raise ConnectionError(f"rendezvous: implausible record length {n}")
    return _recv_exact(sock, n)


class TcpGroup:
    """All ranks of the job joined by TCP connections to rank 0 (star).  ``allgather(blob)`` returns every rank's blob in
    rank order on every rank; ``barrier()`` is an all-gather of nothing.  Collective: every rank makes the same calls."""

    def __init__(self, rank, world, timeout=300.0):
        import socket
        import time
        self.rank, self.world = int(rank), int(world)
        self.peers = {}                 # rank 0: {peer rank: socket}
        self.sock = None                # other ranks: the socket to rank 0
        if self.world == 1:
            return
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        hello = _MAGIC + b"|" + _run_token() + b"|"
        if self.rank == 0:
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
            srv.listen(self.world + 8)
            deadline = time.time() + timeout
            try:
                while len(self.peers) < self.world - 1:
                    srv.settimeout(max(0.1, deadline - time.time()))
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        missing = sorted(set(range(1, self.world)) - set(self.peers))
                        raise TimeoutError(f"rendezvous: ranks {missing} did not join within {timeout:.0f} s") from None
                    try:
                        conn.settimeout(10.0)
                        msg = _recv_msg(conn, 4096)          # the whole greeting, however TCP cut it up
                        if not msg.startswith(hello):
                            raise ValueError("a stranger on our port")
                        peer = int(msg[len(hello):].decode())
                        if not 0 < peer < self.world or peer in self.peers:
                            raise ValueError("rank out of range or seen twice")
                        _send_msg(conn, b"OK")
                        if _recv_msg(conn, 16) != b"ACK":    # the peer has our answer: only now does it count as joined
                            raise ValueError("no acknowledgement")
                        conn.settimeout(timeout)
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self.peers[peer] = conn
                    except (OSError, ValueError, ConnectionError):
                        conn.close()                          # garbled, truncated or foreign greeting: keep listening
            finally:
                srv.close()
            return
        deadline = time.time() + timeout
        while time.time() < deadline:
            for port in _candidate_ports():
                c = None
                try:
                    c = socket.create_connection((addr, port), timeout=2.0)
                    c.settimeout(10.0)
                    _send_msg(c, hello + str(self.rank).encode())
                    if _recv_msg(c, 16) != b"OK":
                        raise ConnectionError("not our listener")
                    _send_msg(c, b"ACK")
                    c.settimeout(timeout)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.sock = c
                    return
                except (OSError, ConnectionError):
                    if c is not None:
                        c.close()
            time.sleep(0.05)
        raise TimeoutError(f"rendezvous: rank {self.rank} found no listener of rank 0 within {timeout:.0f} s")

    def allgather(self, blob):
        blob = bytes(blob)
        if self.world == 1:
            return [blob]
        if self.rank == 0:
            parts = [blob] + [_recv_msg(self.peers[r]) for r in range(1, self.world)]
            packed = b"".join(len(p).to_bytes(8, "little") + p for p in parts)
            for r in range(1, self.world):
                _send_msg(self.peers[r], packed)
            return parts
        _send_msg(self.sock, blob)
        packed = _recv_msg(self.sock)
        parts, o = [], 0
        for _ in range(self.world):
            n = int.from_bytes(packed[o:o + 8], "little")
            parts.append(packed[o + 8:o + 8 + n])
            o += 8 + n
        return parts

    def bcast(self, blob):
        """Rank 0's blob on every rank."""
        return self.allgather(blob if self.rank == 0 else b"")[0]

    def barrier(self):
        self.allgather(b"")

    def close(self):
        for s in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


_group = None


def tcp_group(rank=None, world=None, timeout=300.0):
    """The process-wide TcpGroup (created on first use; collective)."""
    global _group
    if rank is None or world is None:
        rank, world, _ = world_from_env()
    if _group is None or (_group.rank, _group.world) != (int(rank), int(world)):
        _group = TcpGroup(rank, world, timeout)
    return _group


def exchange_bytes(payload, nbytes, rank, world, timeout=300.0):
    """Rank 0 hands ``payload`` (``nbytes`` bytes) to every other rank; returns it.  No third-party module involved (the
    north star's "no PyTorch" covers the rendezvous too)."""
    if world == 1:
        return bytes(payload)
    got = tcp_group(rank, world, timeout).bcast(payload)
    if len(got) != nbytes:
        raise RuntimeError(f"rendezvous: expected {nbytes} bytes from rank 0, got {len(got)}")
    return got


def free_port(addr="127.0.0.1"):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return s.getsockname()[1]


def launch_local(argv, nproc, timeout=None, env_extra=None):
    """Start ``nproc`` copies of ``python argv...`` on this node, one rank per GPU, and wait for them - what
    ``torch.distributed.run`` does for this package, without torch.  The parent must not have touched the GPU (it does
    not here: importing this module initialises nothing).  Each child gets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT and a fresh rendezvous port; every device stays visible in every child (the direct
    all-gather maps its peers' buffers), rank r uses device LOCAL_RANK.  Rank 0's stdout is returned; the other ranks'
    stdout goes to stderr.  Returns (exit code, rank 0's stdout): non-zero as soon as any rank failed, in which case the
    rest are terminated."""
    import subprocess
    import time
    port, rdzv = free_port(), free_port()
    procs = []
    for r in range(nproc):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEIG_RDZV_PORT=str(rdzv),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    deadline = None if timeout is None else time.time() + timeout
    out0, rc, live = b"", 0, set(range(nproc))
    try:
        import threading
        box = {}
        reader = threading.Thread(target=lambda: box.setdefault("out", procs[0].stdout.read()), daemon=True)
        reader.start()
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = code if code > 0 else 1
                        print(f"launch_local: rank {r} exited with status {code}", file=sys.stderr)
            if deadline is not None and time.time() > deadline:
                rc = 124
                print(f"launch_local: timeout after {timeout:.0f} s", file=sys.stderr)
            time.sleep(0.05)
    finally:
        for r in live:                      # a failed or timed-out job: stop exactly the children started here
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        reader.join(5)
        out0 = box.get("out", b"") or b""
    return rc, out0.decode("utf-8", "replace")


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


def gathered_capacity(N, world):
    """Doubles the gathered operand of an N-column operator split over ``world`` ranks can take (upper bound of the
    library's chunk-major layout: slices padded to whole column windows, one scalar line per rank and chunk)."""
    rows = -(-int(N) // int(world))
    return int(world) * (rows + 4 * (1 << 17) + 64) + 4096


def enable_direct_gather(ctx, capacity_doubles, rank=None, world=None):
    """Set up the direct operand exchange (peer writes over every xGMI link at once, csrc/comm_direct.hip) next to the
    RCCL communicator of ``ctx``: every rank allocates its buffers, the hipIpc records travel over the TCP group and
    every rank maps its peers.  Collective.  Returns True when every rank succeeded; on any failure (no peer access, IPC
    refused) all ranks stay on RCCL and the reason is printed on stderr."""
    if rank is None or world is None:
        rank, world, _ = world_from_env()
    grp = tcp_group(rank, world)
    rec, err = b"", ""
    try:
        rec = ctx.direct_alloc(capacity_doubles)
    except Exception as exc:                                    # noqa: BLE001 - reported, every rank falls back together
        err = f"alloc: {exc}"
    recs = grp.allgather(rec)
    ok = all(len(r) == 192 for r in recs)
    if ok:
        try:
            ctx.direct_attach(recs)
        except Exception as exc:                                # noqa: BLE001
            err = f"attach: {exc}"
            ok = False
    oks = grp.allgather(b"1" if ok else b"0")
    if all(o == b"1" for o in oks):
        return True
    if err:
        print(f"[hipeig rank {rank}] direct exchange unavailable ({err}); staying on RCCL", file=sys.stderr)
    try:
        ctx.direct_release()            # a rank that did succeed would keep two gathered operands allocated for nothing
    except Exception:                   # noqa: BLE001 - nothing was allocated
        pass
    return False


def attach_direct(ctx, capacity_doubles, rank=None, world=None):
    """A communicator WITHOUT RCCL for ``ctx``: rank / size from the launcher's environment, the operand exchange and the
    small all-reduces through peer writes (``enable_direct_gather``).  Raises when the peers cannot be mapped - there is
    nothing to fall back to.  Block products on a partitioned operator need RCCL and are refused in this mode."""
    if rank is None or world is None:
        rank, world, _ = world_from_env()
    ctx.attach_direct_only(world, rank)
    if not enable_direct_gather(ctx, capacity_doubles, rank, world):
        raise RuntimeError("the direct exchange could not be set up (see stderr) and this communicator has no RCCL")
    return rank, world


def choose_gather_backend(ctx, H, group, reps=5, trial_wait_s=20.0):
    """Time ``reps`` products of the partitioned operator ``H`` with each exchange backend, check that they give the same
    results (to rounding: the blocked sweep adds a row's terms in no fixed order), and switch every rank to the faster
    one (max over ranks decides, so all ranks agree).  The operand CHANGES from product to product (x, 2x, 3x, ...), so a
    backend that handed the sweep a stale buffer - the previous exchange's data - would be caught, not only one that
    delivers wrong data outright.  The direct backend must have been attached (``enable_direct_gather``).

    The direct backend has never run between DIFFERENT devices (this pool has one GPU per box), so its trial is built
    to fail safely: short wait limits (``trial_wait_s``), NO collective of any kind inside the trial - a rank whose trial
    raises and one whose trial passes still make the same calls - and ONE host-side (TCP) all-gather of (ok, ms) behind
    it; if any rank failed, every rank drains its device, all release the direct buffers together and the run stays on
    RCCL.  Returns a record of what was measured."""
    import numpy as np
    from .hip_vector import HipVector
    n = H.nrows
    tg = tcp_group(ctx.rank, ctx.nranks)
    x = HipVector(np.random.default_rng(4242 + ctx.rank).standard_normal(n), ctx=ctx)
    xs = ctx.alloc(n)
    from . import _lib

    def products(name, out):
        """warm-up, ``reps`` timed products, then 2 * reps compared ones; everything local"""
        ctx.set_gather_backend(name)
        H.apply_shifted(0.0, x._buf, out)                       # warm-up (layout, buffers)
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            H.apply_shifted(0.0, x._buf, out)
        ms = ctx.timer_stop() / reps
        got = []
        for i in range(2 * reps):                               # both buffers of the double-buffered exchange, several times
            _lib.call("hipeig_scale", ctx.handle, n, float(i + 1), x._buf.ptr, xs.ptr)
            H.apply_shifted(0.0, xs, out)
            got.append(HipVector(out).array / (i + 1))          # every product must equal -H x
        return ms, got

    out = ctx.alloc(n)
    group.barrier()
    ms_rccl, ref = products("rccl", out)
    times = {"rccl": group.allmax(ms_rccl)}
    scale = float(np.max(np.abs(ref[0]))) if n else 0.0
    # ---- the direct trial: local only, then one host-side exchange of the verdicts ----
    ok, why, ms_direct, ar_direct = True, "", float("inf"), float("inf")
    tg.barrier()                                                # every rank is past the RCCL part (outside the try: every rank makes it)
    try:
        ctx.set_direct_wait_limit(trial_wait_s)
        ms_direct, got = products("direct", out)
        ok = ctx.gather_info()["wait_error"] == 0
        for a, b in zip(ref, got):
            ok = ok and bool(np.all(np.abs(a - b) <= 1e-12 * scale)) and bool(np.all(np.abs(a - ref[0]) <= 1e-12 * scale))
        if not ok:
            why = "results differ from RCCL's"
        else:
            # the small all-reduce of a MINRES iteration through the mailboxes: a known record ((rank + 1) * (1 .. 16),
            # several rounds so that both mailboxes are used); its peers are exactly the ranks whose exchange just worked
            ctx.set_allreduce_backend("direct")
            want = np.arange(1.0, 17.0) * (ctx.nranks * (ctx.nranks + 1) / 2.0)
            for rnd in range(4):
                rec = HipVector((ctx.rank + 1.0) * np.arange(1.0, 17.0) * (rnd + 1), ctx=ctx)
                ctx.allreduce_vector(rec._buf)
                ok = ok and bool(np.array_equal(rec.array, want * (rnd + 1)))      # small integers: exact in any order
            ar_direct = ctx.allreduce_ms(2, 50)
            if not ok:
                why = "all-reduce through the mailboxes gave a wrong sum"
    except Exception as exc:                                    # noqa: BLE001 - reported, every rank falls back together
        ok, why = False, f"{type(exc).__name__}: {exc}"
    import json
    try:
        ctx.set_exchange(True)                                  # no change of state: drains BOTH streams (pushes run on the
    except Exception:                                           # noqa: BLE001   communication stream), checks no error word
        pass
    verdicts = [json.loads(v.decode()) for v in tg.allgather(json.dumps([bool(ok), float(ms_direct), float(ar_direct), why]).encode())]
    all_ok = all(v[0] for v in verdicts)
    if not all_ok:
        reasons = "; ".join(f"rank {r}: {v[3]}" for r, v in enumerate(verdicts) if not v[0])
        print(f"[hipeig rank {ctx.rank}] direct exchange failed its trial ({reasons}); staying on RCCL", file=sys.stderr)
        try:
            ctx.direct_release()                                # clears the error word with the buffers; every rank has drained
        except Exception:                                       # noqa: BLE001
            pass
        ctx.set_gather_backend("rccl")
        ctx.set_allreduce_backend("rccl")
        group.barrier()
        return {"rccl_ms": round(times["rccl"], 4), "direct_ms": None, "results_agree": False, "direct_trial_failed": reasons,
                "products_compared": 2 * reps, "chosen": "rccl", "reps": reps, "allreduce_chosen": "rccl"}
    ctx.set_direct_wait_limit(0.0)
    times["direct"] = max(v[1] for v in verdicts)
    pick = "direct" if times["direct"] < times["rccl"] else "rccl"
    ctx.set_gather_backend(pick)
    ctx.set_allreduce_backend("rccl")
    ar = {"rccl": group.allmax(ctx.allreduce_ms(2, 50)), "direct": max(v[2] for v in verdicts)}
    pick_ar = "direct" if ar["direct"] < ar["rccl"] else "rccl"
    ctx.set_allreduce_backend(pick_ar)
    if pick == "rccl" and pick_ar == "rccl":
        ctx.direct_release()                                    # nothing uses the two gathered buffers per rank: free them (all ranks decide alike)
    return {"rccl_ms": round(times["rccl"], 4), "direct_ms": round(times["direct"], 4), "results_agree": True,
            "products_compared": 2 * reps, "chosen": pick, "reps": reps, "allreduce_rccl_ms": round(ar["rccl"], 4),
            "allreduce_direct_ms": round(ar["direct"], 4), "allreduce_results_agree": True, "allreduce_chosen": pick_ar}


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

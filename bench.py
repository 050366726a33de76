#!/usr/bin/env python3
"""Headline benchmark: fp64 CSR SpMV with fused shift (y = sigma*x - H x), the kernel that
dominates the inexact-Lanczos shift-and-invert inner loop, at BASELINE.json's metric
configuration (random-sparse Hermitian, N = 1e7, nnz/row ~ 64), plus Lanczos iterations/s
on the same operator.

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: starts its own N ranks, no launcher needed)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W  (a launcher's RANK / WORLD_SIZE are honoured)

One process per GPU.  With N > 1 the SAME N = 1e7 operator is row-partitioned over the ranks
(strong scaling: total work fixed); every step is an RCCL all-gather of the operand slice
followed by the local CSR sweep.  A "step" is one operator application.  Inputs are
generated on the device and are resident in HBM before the timed region.  No torch anywhere:
the launcher's RANK / WORLD_SIZE / MASTER_* variables are read directly, RCCL's unique id
travels over a stdlib TCP exchange and barriers / max-over-ranks use the library's own
all-reduce (eigensolvers_amd.distributed).

Rank 0 prints ONE JSON line; `value` is whole-job algorithmic GB/s (SURVEY.md section 8d
bytes of the GLOBAL operator x steps / max-over-ranks wall time); `roofline` prices the
dominant kernel per GPU against the 8 TB/s HBM peak with HIP events recorded on the
library's own stream; `cpu_baseline` times the same product through scipy's csr_matvec
(the routine the reference's `H @ x` reaches) on a bounded row slab of the same operator.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
_T_START = time.time()

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0   # same guide: measured float4 copy (79 % of spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--size", dest="n", type=int, default=10_000_000)
    ap.add_argument("--nnz-row", type=int, default=64)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--sigma", type=float, default=0.02)
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 CSR-vector, 2 CSR-stream, 3/4 column-window blocked (wave / workgroup units), 5 = 4 with fixed-point accumulators (reproducible)")
    ap.add_argument("--no-lanczos", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-block", action="store_true", help="skip the block-of-8 solve timing (BASELINE config #3, N = 1e6)")
    ap.add_argument("--lanczos-L", type=int, default=8)
    ap.add_argument("--lanczos-maxit", type=int, default=4)
    ap.add_argument("--lanczos-econv", type=float, default=1e-10)
    ap.add_argument("--cpu-lanczos-n", type=int, default=100_000, help="size of the CPU Lanczos baseline instance (~15 s of one core)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, run the TCP rendezvous and an all-gather of the rank ids, print the JSON line, touch no GPU")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="self-spawned ranks are stopped after this many seconds")
    return ap.parse_args()


def kernel_signature(layout):
    """What a counter profile of the sweep is valid for: the sources of the sweep kernels (device code and layout builder)
    and the layout constants of the operator copy they ran on.  `profiles/pmc_current.json` carries the signature of
    the run it was collected from."""
    import hashlib
    h = hashlib.sha256()
    for f in ("spmv_device.h", "spmv.hip"):
        h.update(open(os.path.join(REPO, "eigensolvers_amd", "csrc", f), "rb").read())
    h.update(json.dumps(layout, sort_keys=True).encode())
    return h.hexdigest()[:16]


def global_bytes(N, nnz):
    return nnz * 12 + (N + 1) * 4 + 8 * N + 8 * N


def rendezvous_only(a, result):
    """The multi-rank start-up without a GPU: every rank joins the TCP group, hands its rank id round and checks what
    it got back; rank 0 prints the line.  What `pytest -m "not gpu"` runs to see `bench.py --gpus 2` start plainly."""
    from eigensolvers_amd import distributed as D
    rank, world, local_rank = D.world_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    grp = D.tcp_group(rank, world, timeout=60.0)
    seen = [int(b.decode()) for b in grp.allgather(str(rank).encode())]
    uid = D.exchange_bytes(bytes(range(128)) if rank == 0 else b"", 128, rank, world)
    grp.barrier()
    ok = seen == list(range(world)) and uid == bytes(range(128))
    if rank == 0:
        result["line"] = json.dumps({"rendezvous": "ok" if ok else "failed", "n_ranks_seen": len(seen), "n_gpus": world,
                                     "launcher": os.environ.get("HIPEIG_LAUNCHER", "external")})
    if not ok:
        raise SystemExit(3)


def main(result):
    a = parse()
    if a.rendezvous_only:
        return rendezvous_only(a, result)
    import ctypes as C
    import numpy as np
    import eigensolvers_amd as ea
    from eigensolvers_amd import distributed as D

    rank, world, local_rank = D.world_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    force = os.environ.get("HIPEIG_FORCE_COLLECTIVES", "0") not in ("", "0")     # rehearse the RCCL path on one rank
    try:
        ctx = ea.HipContext(local_rank)
    except ea._lib.HipEigError:                      # launcher already narrowed the visible devices to one
        ctx = ea.HipContext(0)
    ea.HipContext._default = ctx
    # HIPEIG_COMM: "auto" (default) = RCCL communicator + the direct peer-write exchange next to it, the faster of the two
    # is picked per operator by timing; "rccl" = RCCL only; "direct" = no RCCL at all (also what lets several ranks
    # share ONE GPU, which RCCL refuses - a rehearsal of the whole multi-rank run on a single-GPU box)
    comm_mode = os.environ.get("HIPEIG_COMM", "auto")
    rccl_lib, direct_ok = None, False
    if world > 1 or force:
        cap = D.gathered_capacity(a.n, world)
        if comm_mode == "direct":
            D.attach_direct(ctx, cap, rank, world)
            direct_ok = True
        else:
            D.attach_rccl(ctx, rank, world)
            buf = C.create_string_buffer(512)
            ea._lib.call("hipeig_comm_library", buf, 512)
            rccl_lib = buf.value.decode()
            if comm_mode == "auto" and world > 1:
                direct_ok = D.enable_direct_gather(ctx, cap, rank, world)
    group = D.DeviceGroup(ctx)
    barrier, allmax, allsum = group.barrier, group.allmax, group.allsum
    n_ranks_seen = int(round(allsum(1.0)))           # one contribution per rank through the communicator itself

    N = a.n
    b, e = D.row_range(N, world, rank)
    x = ea.HipVector(np.random.default_rng(100 + rank).standard_normal(e - b), ctx=ctx)
    y = ctx.alloc(e - b)

    def make_operator():
        t0 = time.time()
        Hn = ea.HipCsrOperator.generate(N, a.nnz_row, seed=a.seed, row_begin=b, row_end=e, ctx=ctx)
        if a.variant:
            Hn.set_variant(a.variant)
        barrier()
        return Hn, time.time() - t0

    def pick_exchange(Hn):
        if direct_ok and comm_mode == "auto":
            return D.choose_gather_backend(ctx, Hn, group, reps=5)         # RCCL vs peer writes, by timing 5 products of each
        gi = ctx.gather_info()
        return {"chosen": gi["backend"], "allreduce_chosen": gi["allreduce_backend"],
                "why": "HIPEIG_COMM=" + comm_mode if comm_mode != "auto" else "direct exchange unavailable (see stderr)"}

    exchange = None
    if world > 1 and os.environ.get("HIPEIG_BENCH_CHUNK_TRIALS", "1") not in ("", "0") and "HIPEIG_GATHER_CHUNKS" not in os.environ:
        # How many chunks the operand exchange is cut into decides between fewer launches / slabs (one chunk: a rank's
        # sweeps alone 0.30-0.34 ms at 8 ranks) and more overlap (two: 0.33-0.35 ms, but the first half of the remote
        # windows starts when half the exchange is through) - which wins depends on the exchange time of the machine, so it
        # is measured: build the operator with one and with two chunks (~1 s each), let each pick its exchange backend, time
        # 10 products, keep the faster and rebuild it.  Only the timings of the trials are kept, never two operators at once.
        # HIPEIG_BENCH_CHUNK_TRIALS=0 takes the library's layout rule instead (one build).
        tried = {}
        for nch in (1, 2):
            ctx.set_gather_chunks(nch)
            Hn, _ = make_operator()
            pick_exchange(Hn)
            for _ in range(3):
                Hn.apply_shifted(a.sigma, x._buf, y)
            barrier()
            ctx.timer_start()
            for _ in range(10):
                Hn.apply_shifted(a.sigma, x._buf, y)
            tried[nch] = allmax(ctx.timer_stop() / 10)
            del Hn
            ctx.synchronize()
        nch_best = min(tried, key=tried.get)
        ctx.set_gather_chunks(nch_best)
        H, t_gen = make_operator()
        exchange = dict(pick_exchange(H), chunks_tried={str(k): round(v, 4) for k, v in tried.items()}, chunks_chosen=nch_best)
    else:
        # The number of exchange chunks follows from the layout (the library's rule: two when a rank's slice spans >= 8
        # column windows, so that the first chunk's windows are swept while the second travels; HIPEIG_GATHER_CHUNKS
        # overrides); ONE operator build per rank, then only the exchange backends are timed (5 + 10 products each).
        H, t_gen = make_operator()
        if world > 1:
            exchange = pick_exchange(H)
    nnz_total = int(allsum(float(H.nnz)))

    startup_s = time.time() - _T_START                # process start -> first warm-up product (imports, rendezvous, build, backend choice)
    for _ in range(a.warmup):
        H.apply_shifted(a.sigma, x._buf, y)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(a.steps):
        H.apply_shifted(a.sigma, x._buf, y)
    dev_ms = ctx.timer_stop()              # HIP events on the stream the kernels run on
    barrier()
    wall = allmax(time.perf_counter() - t0)
    dev_ms = allmax(dev_ms)
    # spread of single steps (median + min), measured AFTER the timed region, one event pair per step
    singles = []
    for _ in range(min(a.steps, 20)):
        ctx.timer_start()
        H.apply_shifted(a.sigma, x._buf, y)
        singles.append(ctx.timer_stop())
    singles.sort()
    step_median, step_min = allmax(singles[len(singles) // 2]), allmax(singles[0])
    # where a partitioned product spends its time (events on both streams, max over ranks of the per-rank averages) and
    # what one small all-reduce of a MINRES iteration costs
    phases = None
    if world > 1 or force:
        ctx.phase_timing(True)
        acc = {}
        for _ in range(10):
            H.apply_shifted(a.sigma, x._buf, y)
            for k, v in ctx.phase_times().items():
                if v is not None:
                    acc.setdefault(k, []).append(v)
        ctx.phase_timing(False)
        phases = {}
        for k in ("gather_ms", "local_sweep_ms", "remote_sweep_ms", "product_ms", "idle_before_first_chunk_ms"):
            v = allmax(sum(acc[k]) / len(acc[k]) if k in acc else -1.0)     # the same collective calls on every rank
            phases[k] = round(v, 5) if v >= 0 else None
        phases["allreduce_ms"] = round(allmax(ctx.allreduce_ms(2, 50)), 5)
        phases["exchange_chunks"] = H.layout_info()["exchange_chunks"]
        # One rank's sweeps ALONE on its device, exchange switched off (own slice placed, peers' parts stale): the compute
        # part of the product.  On a real node every rank has its own GPU and this equals local + remote sweep; in a
        # rehearsal with several ranks on ONE GPU the ranks take turns (host barrier between them), which separates the
        # cost of the split sweep from the cost of sharing the GPU.
        if world > 1:
            tg = D.tcp_group(rank, world)
            alone = -1.0
            for r in range(world):
                tg.barrier()
                if r == rank:
                    ctx.set_exchange(False)
                    for _ in range(2):
                        H.apply_shifted(a.sigma, x._buf, y)
                    ctx.timer_start()
                    for _ in range(10):
                        H.apply_shifted(a.sigma, x._buf, y)
                    alone = ctx.timer_stop() / 10
                    ctx.set_exchange(True)
            tg.barrier()
            phases["sweeps_alone_ms"] = round(allmax(alone), 5)

    # what a plain streaming kernel reaches on THIS device (confirm the nominal figure on the
    # box): out-of-place scale of 1e8 doubles, 0.8 GB read + 0.8 GB written per call, well past the Infinity Cache
    ncopy = 100_000_000
    ca, cb = ctx.alloc(ncopy), ctx.alloc(ncopy)
    ea._lib.call("hipeig_vec_fill", ctx.handle, ca.ptr, ncopy, 1.0)
    ea._lib.call("hipeig_scale", ctx.handle, ncopy, 1.0000001, ca.ptr, cb.ptr)
    ctx.timer_start()
    for _ in range(10):
        ea._lib.call("hipeig_scale", ctx.handle, ncopy, 1.0000001, ca.ptr, cb.ptr)
    copy_gbs = -allmax(-(16.0 * ncopy / 1e9) / (ctx.timer_stop() / 10 / 1e3))     # the slowest rank's rate
    del ca, cb

    gbytes = global_bytes(N, nnz_total) / 1e9
    value = gbytes * a.steps / wall
    per_gpu_bytes = H.algorithmic_bytes() / 1e9          # local rows, x counted once (full length)
    achieved = per_gpu_bytes / (dev_ms / a.steps / 1e3)
    nlaunch = H.launches_per_apply()                     # the blocked layouts sweep the rows in 1-2 launches
    # HBM traffic per launch from the committed PMC pass (cannot be collected inside this process)
    traffic, kname = None, {"csr-vector": "spmv_vector_kernel", "csr-stream": "spmv_stream_kernel",
                            "column-window-blocked(wave)": "spmv_tcoo_kernel",
                            "column-window-blocked(workgroup)": "spmv_tcoow_kernel",
                            "column-window-blocked(workgroup, fixed-point)": "spmv_tcoow_kernel"}[H.last_variant()]
    signature = kernel_signature(H.layout_info())
    traffic_note = None
    try:
        pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_current.json")))
        if pmc["config"] != {"N": N, "nnz_row": a.nnz_row, "n_gpus": world}:
            traffic_note = "no counter pass for this configuration"
        elif pmc.get("signature") != signature:
            # the kernel or its layout changed since the counters were collected: do not report stale traffic
            traffic_note = f"stale: counters were collected for kernel signature {pmc.get('signature')}, this run is {signature}"
        else:
            hits = [v for k, v in pmc["kernels"].items() if kname in k and "hbm_bytes_per_launch" in v]      # rocprofv3 names carry "void", template arguments
            traffic = hits[0]["hbm_bytes_per_launch"] if hits else None
            if traffic is None:
                traffic_note = f"the counter pass holds no kernel named like {kname}"
    except Exception as exc:
        traffic, traffic_note = None, f"profiles/pmc_current.json unreadable: {exc}"
    out = {
        "metric": "fp64 CSR SpMV GB/s (fused shift y = sigma*x - H x)", "value": round(value, 2), "unit": "GB/s",
        "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(wall / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"random-sparse Hermitian CSR N={N} nnz/row~{a.nnz_row} (BASELINE metric config), "
                               f"row-partitioned over {world} GPU(s)",
                   "N": N, "nnz": nnz_total, "nnz_per_row": round(nnz_total / N, 3), "nnz_row_arg": a.nnz_row, "sigma": a.sigma,
                   "kernel_variant": ("auto:" if a.variant == 0 else "forced:") + H.last_variant(),
                   "generator_seed": a.seed, "generate_s": round(t_gen, 2), "startup_s": round(allmax(startup_s), 2),
                   "rccl_library": rccl_lib,
                   "comm": comm_mode, "exchange": exchange, "layout": H.layout_info()},
        "roofline": {"bound": "hbm", "kernel": kname,
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 4), "achievable_peak": HBM_ACHIEVABLE_GBS,
                     "measured_stream_GBs": round(copy_gbs, 1), "frac_of_measured_stream": round(achieved / copy_gbs, 4),
                     "traffic": traffic,
                     "traffic_source": "profiles/pmc_current.json (separate rocprofv3 --pmc passes)" if traffic else traffic_note,
                     "kernel_signature": signature,
                     "launches_per_step": nlaunch,
                     "algorithmic_bytes_per_launch": int(H.algorithmic_bytes() // nlaunch),
                     "avg_launch_ms": round(dev_ms / a.steps / nlaunch, 5),
                     "ms_per_step_device": round(dev_ms / a.steps, 5),
                     "single_step_ms_median": round(step_median, 5), "single_step_ms_min": round(step_min, 5)},
    }
    if phases is not None:
        out["phases"] = phases

    # ---- Lanczos iterations/s on the same operator (outside the timed SpMV region) ----
    if not a.no_lanczos:
        opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}}
        from eigensolvers_amd.generators import guess_vector
        v0 = ea.HipVector(guess_vector(N, 1, b, e).copy(), opts, ctx=ctx)
        barrier()
        tl = time.perf_counter()
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):          # keep stdout to the single JSON line
            ev, Y, st = ea.inexactLanczosDiagonalization(H, v0, a.sigma, a.lanczos_L, a.lanczos_maxit, a.lanczos_econv,
                                                         writeOut=False)
        barrier()
        tl = allmax(time.perf_counter() - tl)
        res = ea.true_residual_norms(H, ev, Y, 1)
        inner = Y[0].last_solve_stats["iterations"] if getattr(Y[0], "last_solve_stats", None) else None
        if inner is None and v0.last_solve_stats:
            inner = v0.last_solve_stats["iterations"]
        out["lanczos"] = {"cum_iters": st["cumIter"], "seconds": round(tl, 3),
                          "iters_per_s": round(st["cumIter"] / tl, 4), "ritz_value": float(ev[0]),
                          "converged": bool(st["isConverged"]), "eigenvalue_change_residual": float(st["residual"]),
                          "true_residual_norm": float(res[0]), "minres_iters_last_solve": inner,
                          "L": a.lanczos_L, "maxit": a.lanczos_maxit, "eConv": a.lanczos_econv, "linear_tol": 1e-10}

    # ---- BASELINE config #3 (N = 1e6, 32 nnz/row, block of 8): lock-step block solve vs eight single solves ----
    if world == 1 and not a.no_block:
        Nb = 1_000_000
        Hb = ea.HipCsrOperator.generate(Nb, 32, seed=a.seed, ctx=ctx)
        Qb = np.random.default_rng(5).standard_normal((Nb, 8))
        Qb /= np.linalg.norm(Qb, axis=0)
        ob = lambda: {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 4000, "linear_tol": 1e-10}}
        Xb = [ea.HipVector(Qb[:, j].copy(), ob(), ctx=ctx) for j in range(8)]
        ea.HipVector.solveBlock(Hb, Xb, a.sigma)                          # builds the block layout, warms up
        yb = ctx.alloc(Nb)
        Hb.apply_shifted(a.sigma, Xb[0]._buf, yb)                         # builds the single-vector layout too: both timings
        del yb                                                            # below are of solves, not of a first-use layout build
        ctx.synchronize()
        tb = time.perf_counter()
        Wb = ea.HipVector.solveBlock(Hb, Xb, a.sigma)
        ctx.synchronize()
        tb = time.perf_counter() - tb
        ts = time.perf_counter()
        Sb = [ea.HipVector.solve(Hb, xb, a.sigma) for xb in Xb]
        ctx.synchronize()
        ts = time.perf_counter() - ts
        itb = [w.last_solve_stats["iterations"] for w in Wb]
        out["block8"] = {"N": Nb, "nnz": int(Hb.nnz), "kernel": Hb.block_info()["variant"], "iterations": itb,
                         "iterations_equal_single": itb == [w.last_solve_stats["iterations"] for w in Sb],
                         "block_seconds": round(tb, 4), "single_seconds": round(ts, 4),
                         "solves_per_s": round(8 / tb, 2), "single_solves_per_s": round(8 / ts, 2),
                         "ms_per_block_iteration": round(1e3 * tb / max(itb), 4), "speedup": round(ts / tb, 3),
                         "max_rel_diff": float(max(ea.HipVector.linearCombination([Wb[j], Sb[j]], [1.0, -1.0]).norm() / Sb[j].norm()
                                                   for j in range(8)))}
        # one MINRES iteration on the SAME N = 1e6 operator, device and (below) CPU restatement of the reference path
        out["minres_n1e6"] = {"gpu_ms_per_iteration": round(1e3 * ts / max(1, sum(w.last_solve_stats["iterations"] for w in Sb)), 5),
                              "iterations_per_solve": Sb[0].last_solve_stats["iterations"], "N": Nb, "nnz": int(Hb.nnz)}
        Hb_host = Hb.to_scipy() if (rank == 0 and not a.no_cpu) else None
        b_host = Qb[:, 0].copy()
        del Hb, Xb, Wb, Sb

    # ---- CPU baseline: scipy csr_matvec on a bounded row slab (rank 0, single GPU only) ----
    if rank == 0 and world == 1 and not a.no_cpu:
        rows = min(N, max(1000, int(2.0e7 // max(a.nnz_row, 1))))     # ~2e7 non-zeros
        slab = ea.HipCsrOperator.generate(N, a.nnz_row, seed=a.seed, row_begin=0, row_end=rows, ctx=ctx).to_scipy()
        xh = np.random.default_rng(100).standard_normal(N)
        slab @ xh
        reps, tc = 0, time.perf_counter()
        while reps < 200 and time.perf_counter() - tc < 12.0:
            yh = a.sigma * xh[:rows] - slab @ xh
            reps += 1
        tc = time.perf_counter() - tc
        sb = (slab.nnz * 12 + (rows + 1) * 4 + 8 * N + 8 * rows) / 1e9
        try:
            import threadpoolctl
            blas = [(d.get("internal_api"), d.get("num_threads")) for d in threadpoolctl.threadpool_info()]
        except Exception:
            blas = None
        # (ii) Lanczos iterations/s of the CPU restatement of the reference path (oracle/: NumPy + SciPy MINRES,
        # numpyVector.py:147-178 under inexact_Lanczos.py:229-443) on a bounded instance of the same generator
        from oracle import lanczos_ref
        from oracle.numpy_vector import RefVector
        from eigensolvers_amd.generators import gapped_csr_host, guess_vector
        n_cpu = min(N, a.cpu_lanczos_n)
        Hc = gapped_csr_host(n_cpu, a.nnz_row, seed=a.seed)
        tl0 = time.perf_counter()
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):       # the oracle's "not converged" alert belongs in the JSON, not on stderr
            evc, Yc, stc = lanczos_ref.inexact_lanczos(Hc, RefVector(guess_vector(n_cpu, 1).copy(), {"linearSystemArgs": {
                "linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}}), a.sigma, 3, 1, a.lanczos_econv)
        tl0 = time.perf_counter() - tl0
        cpu_lanczos = {"N": n_cpu, "nnz_per_row": round(Hc.nnz / n_cpu, 2), "cum_iters": int(stc["cumIter"]),
                       "seconds": round(tl0, 2), "iters_per_s": round(stc["cumIter"] / tl0, 4), "ritz_value": float(evc[0]),
                       "converged": bool(stc["isConverged"]),
                       "what": "oracle.lanczos_ref (NumPy/SciPy restatement of the reference path), ONE cycle of L=3 (maxit = 1: "
                               "a bounded sample, not run to convergence), minres rtol 1e-10, single-threaded csr_matvec"}
        # (iii) the same inputs as the device: MINRES iterations of the oracle on the N = 1e6 config-#2 operator
        # (numpyVector.py:163 -> scipy minres; here oracle.minres_ref, the restatement the parity tests use)
        cpu_minres = None
        if "minres_n1e6" in out and Hb_host is not None:
            from oracle import minres_ref
            nit = 8
            t0c = time.perf_counter()
            minres_ref.minres(lambda v: a.sigma * v - Hb_host @ v, b_host, rtol=1e-30, maxiter=nit)
            t0c = time.perf_counter() - t0c
            cpu_minres = {"cpu_ms_per_iteration": round(1e3 * t0c / nit, 3), "iterations_timed": nit, "cores": 1,
                          "N": out["minres_n1e6"]["N"], "nnz": out["minres_n1e6"]["nnz"],
                          "gpu_ms_per_iteration": out["minres_n1e6"]["gpu_ms_per_iteration"],
                          "what": "oracle.minres_ref on the same N = 1e6, 32 nnz/row operator and right-hand side as the device solve"}
            del Hb_host
        out["cpu_baseline"] = {"value": round(sb * reps / tc, 3), "unit": "GB/s", "cores": 1, "kind": "port",
                               "lanczos": cpu_lanczos, "minres_same_inputs": cpu_minres,
                               "sample": f"rows [0,{rows}) of the same operator ({slab.nnz} nnz), x of full length {N}, "
                                         f"{reps} reps of sigma*x - H@x via scipy.sparse csr_matvec (single-threaded), "
                                         f"{tc:.1f} s",
                               "host_cpus": os.cpu_count(), "blas_threads": blas,
                               "numpy": np.__version__, "scipy": __import__("scipy").__version__}
    if rank == 0:
        result["line"] = json.dumps(out)
    barrier()


if __name__ == "__main__":
    _a = parse()
    if _a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: this process becomes the launcher - it spawns one rank per GPU BEFORE anything here has
        # touched a GPU, relays rank 0's JSON line and exits non-zero if any rank did
        from eigensolvers_amd.distributed import launch_local
        _rc, _out = launch_local([os.path.abspath(__file__)] + sys.argv[1:], _a.gpus, timeout=_a.launch_timeout,
                                 env_extra={"HIPEIG_LAUNCHER": "bench.py"})
        _lines = [ln for ln in _out.splitlines() if ln.startswith("{")]
        if _lines:
            print(_lines[-1], flush=True)
        sys.exit(_rc if _rc else (0 if _lines else 5))
    # Native libraries (gloo, RCCL) write banners to file descriptor 1; keep stdout to the one JSON
    # line by running everything with fd 1 pointed at stderr and printing the result afterwards.
    from eigensolvers_amd.distributed import stdout_to_stderr
    _result = {}
    with stdout_to_stderr():
        main(_result)
    if "line" in _result:
        print(_result["line"], flush=True)

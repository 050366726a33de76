"""The row-partitioned multi-rank path rehearsed on ONE GPU: P contexts in this process, one host
thread per rank, joined by the library's in-process loopback collectives (hipeig.h,
``hipeig_comm_init_loopback``).  Everything a P-GPU run does except RCCL's own transport is
exercised with real data from several ranks: ragged row ranges and the padded all-gather stride,
the column remap of every kernel layout, the split (local windows / remote windows) sweep with
rank > 0, reductions assembled from several ranks' partial sums, MINRES and the Lanczos driver
taking identical decisions on every rank.  RCCL itself is covered on a one-rank communicator by
test_gpu_collectives.py; real multi-GPU runs are the driver's scaling bench."""
import numpy as np
import pytest

from conftest import load_golden
from eigensolvers_amd.distributed import LoopbackGroup, row_range
from eigensolvers_amd.generators import guess_vector

pytestmark = pytest.mark.gpu

OPTS = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-10}}


@pytest.mark.parametrize("P", [2, 3])
def test_partitioned_lanczos_on_loopback_ranks(hip, gapped4000, P):
    Hh, guess = gapped4000
    N = 4000
    x = np.random.default_rng(3).standard_normal(N)
    y_ref = Hh @ x
    g = load_golden("gapped_csr_n4000_minres.npz")
    grp = LoopbackGroup(P)

    def body(rank, ctx):
        b, e = row_range(N, P, rank)
        H = hip.HipCsrOperator.generate(N, 32, seed=7, row_begin=b, row_end=e, ctx=ctx)
        out = {"range": (b, e)}
        for variant in (1, 2, 3, 4):
            H.set_variant(variant)
            y = hip.HipVector(x[b:e], ctx=ctx).applyOp(H).array
            out[f"spmv{variant}"] = float(np.max(np.abs(y - y_ref[b:e])))
        H.set_variant(0)
        X = hip.HipVector(x[b:e], ctx=ctx)
        out["dot"] = X.vdot(X)
        out["norm"] = X.norm()
        Ys = [hip.HipVector(np.random.default_rng(10 + i).standard_normal(N)[b:e], ctx=ctx) for i in range(3)]
        out["gram"] = hip.HipVector.overlapMatrix(Ys)
        v0 = hip.HipVector(guess[b:e].copy(), dict(OPTS), ctx=ctx)
        ev, Y, st = hip.inexactLanczosDiagonalization(H, v0, 0.02, 8, 10, 1e-13, writeOut=False)
        out.update(ev=ev, cumIter=st["cumIter"], conv=st["isConverged"], y0=Y[0].array,
                   its=Y[0].last_solve_stats, res=hip.true_residual_norms(H, ev, Y, 1)[0])
        return out

    try:
        res = grp.run(body)
    finally:
        grp.close()
    full = [np.random.default_rng(10 + i).standard_normal(N) for i in range(3)]
    G = np.array([[np.dot(a, b) for b in full] for a in full])
    for r, o in enumerate(res):
        assert o["range"] == row_range(N, P, r)
        for variant in (1, 2, 3, 4):
            assert o[f"spmv{variant}"] < 1e-12, (r, variant, o[f"spmv{variant}"])
        assert abs(o["dot"] - np.dot(x, x)) < 1e-10 * np.dot(x, x)
        assert abs(o["norm"] - np.linalg.norm(x)) < 1e-12 * np.linalg.norm(x)
        np.testing.assert_allclose(o["gram"], G, rtol=0, atol=1e-10)
        # every rank holds the same scalars, bit for bit, and took the same decisions
        np.testing.assert_array_equal(o["ev"], res[0]["ev"])
        assert o["dot"] == res[0]["dot"] and o["cumIter"] == res[0]["cumIter"]
        assert abs(o["ev"][0] - g["ev"][0]) <= 1e-10 * abs(g["ev"][0])          # north-star tolerance
        assert o["cumIter"] == int(g["cumIter"]) and o["conv"]
        assert o["res"] < 1e-6
    y0 = np.concatenate([o["y0"] for o in res])
    assert y0.shape == (N,) and abs(abs(np.dot(y0, g["vec0"])) - 1) < 1e-8


@pytest.mark.parametrize("csplit", [None, 3])
def test_split_sweep_and_minres_with_two_ranks(hip, csplit, monkeypatch):
    """Three column windows over two ranks: rank 0 owns window 0, rank 1 owns window 2, window 1
    straddles both - each rank sweeps its own window under the all-gather and the rest after it.
    csplit = 3: additionally three workgroups per row block (column splits), i.e. the local and the
    remote launch both leave raw slabs that a combine launch adds up - the many-GPU configuration."""
    N, P = 300_000, 2
    if csplit:
        monkeypatch.setenv("HIPEIG_TCOOW_CSPLIT", str(csplit))
    single = hip.HipCsrOperator.generate(N, 32, seed=5)
    single.set_variant(2)
    x = np.random.default_rng(4).standard_normal(N)
    y_ref = hip.HipVector(x).applyOp(single).array
    b_full = guess_vector(N, 2) / np.linalg.norm(guess_vector(N, 2))
    w_ref = hip.HipVector.solve(single, hip.HipVector(b_full.copy(), dict(OPTS)), 0.02)
    it_ref, w_ref = w_ref.last_solve_stats["iterations"], w_ref.array
    grp = LoopbackGroup(P)

    def body(rank, ctx):
        b, e = row_range(N, P, rank)
        H = hip.HipCsrOperator.generate(N, 32, seed=5, row_begin=b, row_end=e, ctx=ctx)
        out = {}
        for variant in (2, 4):
            H.set_variant(variant)
            y = hip.HipVector(x[b:e], ctx=ctx).applyOp(H).array
            out[f"spmv{variant}"] = float(np.max(np.abs(y - y_ref[b:e])) / np.max(np.abs(y_ref)))
            w = hip.HipVector.solve(H, hip.HipVector(b_full[b:e].copy(), dict(OPTS), ctx=ctx), 0.02)
            out[f"it{variant}"] = w.last_solve_stats["iterations"]
            out[f"coll{variant}"] = w.last_solve_stats["collectives"]
            out[f"w{variant}"] = w.array
        return out

    try:
        res = grp.run(body)
    finally:
        grp.close()
    for variant in (2, 4):
        for o in res:
            assert o[f"spmv{variant}"] < 1e-14
            assert abs(o[f"it{variant}"] - it_ref) <= 2
            assert o[f"it{variant}"] == res[0][f"it{variant}"]                 # identical decisions on every rank
            # per iteration ONE operand exchange (it carries every rank's share of <y,y>) and ONE all-reduce
            # (<v,y> + the lagged <x,x>); nothing else - iterations are enqueued in chunks of 16 and the stop of
            # iteration K is seen in KC of iteration K + 1, so (K + 1) iterations rounded up to whole chunks were issued
            enq = 16 * -(-(o[f"it{variant}"] + 1) // 16)
            assert o[f"coll{variant}"] == 2 * enq, (o[f"coll{variant}"], enq)
        w = np.concatenate([o[f"w{variant}"] for o in res])
        assert np.linalg.norm(w - w_ref) <= 1e-8 * np.linalg.norm(w_ref)


@pytest.mark.parametrize("chunks", [1, 2, 3])
def test_chunked_exchange_layout_on_three_ranks(hip, chunks, monkeypatch):
    """The chunk-major gathered operand: every rank's slice cut into `chunks` pieces, chunk c of
    all ranks contiguous, one collective per chunk, the sweep of a chunk's column windows behind that chunk's arrival.
    Ragged slabs (N not divisible by 3), several windows per chunk, with and without column splits; products of every
    kernel layout, MINRES and the block product must not notice the layout."""
    N, P = 250_001, 3
    monkeypatch.setenv("HIPEIG_GATHER_CHUNKS", str(chunks))
    monkeypatch.setenv("HIPEIG_TCOOW_WBITS", "13")                      # 8 Ki-column windows: ~10 per rank slice
    single = hip.HipCsrOperator.generate(N, 24, seed=21)
    single.set_variant(2)
    x = np.random.default_rng(5).standard_normal(N)
    y_ref = hip.HipVector(x).applyOp(single).array
    b_full = guess_vector(N, 4) / np.linalg.norm(guess_vector(N, 4))
    w_ref = hip.HipVector.solve(single, hip.HipVector(b_full.copy(), dict(OPTS)), 0.02)
    it_ref, w_ref = w_ref.last_solve_stats["iterations"], w_ref.array
    Xh = np.random.default_rng(6).standard_normal((N, 3))
    yb_ref = [hip.HipVector(Xh[:, j].copy()).applyOp(single).array for j in range(3)]

    def run(csplit):
        if csplit:
            monkeypatch.setenv("HIPEIG_TCOOW_CSPLIT", str(csplit))
        else:
            monkeypatch.delenv("HIPEIG_TCOOW_CSPLIT", raising=False)
        grp = LoopbackGroup(P)

        def body(rank, ctx):
            b, e = row_range(N, P, rank)
            H = hip.HipCsrOperator.generate(N, 24, seed=21, row_begin=b, row_end=e, ctx=ctx)
            out = {}
            for variant in (1, 2, 3, 4, 5):
                H.set_variant(variant)
                y = hip.HipVector(x[b:e], ctx=ctx).applyOp(H).array
                out[f"spmv{variant}"] = float(np.max(np.abs(y - y_ref[b:e])) / np.max(np.abs(y_ref)))
            H.set_variant(4)
            out["layout"] = H.layout_info()
            w = hip.HipVector.solve(H, hip.HipVector(b_full[b:e].copy(), dict(OPTS), ctx=ctx), 0.02)
            out["it"], out["coll"], out["w"] = w.last_solve_stats["iterations"], w.last_solve_stats["collectives"], w.array
            Y = H.apply_block([hip.HipVector(Xh[b:e, j].copy(), ctx=ctx)._buf for j in range(3)])
            out["yb"] = [hip.HipVector(yy).array for yy in Y]
            return out

        try:
            return grp.run(body)
        finally:
            grp.close()

    for csplit in (None, 2):
        res = run(csplit)
        for o in res:
            assert o["layout"]["exchange_chunks"] == chunks and o["layout"]["column_splits"] == (csplit or 1)
            for variant in (1, 2, 3, 4):
                assert o[f"spmv{variant}"] < 1e-14, (variant, o[f"spmv{variant}"])
            assert o["spmv5"] < 1e-12                                    # fixed-point accumulators: absolute error bound
            assert o["it"] == res[0]["it"] and abs(o["it"] - it_ref) <= 2
            assert o["coll"] == 2 * 16 * -(-(o["it"] + 1) // 16)         # two collectives per iteration whatever the chunking
        w = np.concatenate([o["w"] for o in res])
        assert np.linalg.norm(w - w_ref) <= 1e-8 * np.linalg.norm(w_ref)
        for j in range(3):
            yb = np.concatenate([o["yb"][j] for o in res])
            assert np.max(np.abs(yb - yb_ref[j])) <= 1e-13 * np.max(np.abs(yb_ref[j]))


@pytest.mark.parametrize("rtol", [1e-6, 1e-10])
def test_partitioned_minres_on_a_near_eigenvector(hip, rtol):
    """a right-hand side that is an eigenvector + 1e-9 noise (what restarts and converged Ritz vectors
    hand to the inner solve).  Round 2's partitioned path took beta^2 as <y,y> - alfa^2, which cancels here (wrong by
    50 % at 1e-8, clamped to 0 -> NaN at 1e-9); now every rank's share of the directly reduced <y,y> rides on the
    operand exchange, so the partitioned solve does what the single-GPU solve and the oracle do: same iteration count,
    same stop code, same solution - single right-hand side and lock-step block."""
    import scipy.sparse.linalg as spl
    from oracle import minres_ref
    from eigensolvers_amd.generators import gapped_csr_host
    N, P, sigma = 6000, 3, 0.02
    Hh = gapped_csr_host(N, 32, seed=7)
    lam, vec = spl.eigsh(Hh, k=1, sigma=sigma, which="LM")
    rng = np.random.default_rng(17)
    rhs = []
    for eps in (1e-8, 1e-9, 1e-10):
        b = vec[:, 0] + eps * rng.standard_normal(N)
        rhs.append(b / np.linalg.norm(b))
    opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 500, "linear_tol": rtol}}
    single = hip.HipCsrOperator.from_scipy(Hh)
    ref = []
    for b in rhs:
        w = hip.HipVector.solve(single, hip.HipVector(b.copy(), dict(opts)), sigma)
        xo, info, itn, istop = minres_ref.minres(lambda v: sigma * v - Hh @ v, b, rtol=rtol, maxiter=500)
        assert (w.last_solve_stats["iterations"], w.last_solve_stats["istop"]) == (itn, istop) and info == 0
        ref.append((itn, istop, w.array))
    grp = LoopbackGroup(P)

    def body(rank, ctx):
        lo, hi = row_range(N, P, rank)
        H = hip.HipCsrOperator.from_scipy(Hh, row_begin=lo, row_end=hi, ctx=ctx)
        out = []
        for b in rhs:
            w = hip.HipVector.solve(H, hip.HipVector(b[lo:hi].copy(), dict(opts), ctx=ctx), sigma)
            out.append((w.last_solve_stats["iterations"], w.last_solve_stats["istop"], w.array))
        W = hip.HipVector.solveBlock(H, [hip.HipVector(b[lo:hi].copy(), dict(opts), ctx=ctx) for b in rhs], sigma)
        blk = [(w.last_solve_stats["iterations"], w.last_solve_stats["istop"], w.array) for w in W]
        return out, blk

    try:
        res = grp.run(body)
    finally:
        grp.close()
    for j, (itn, istop, w_ref) in enumerate(ref):
        if rtol == 1e-6:
            assert itn <= 3                                              # over in a step or two: beta_2 is ~1e-8 of alfa_1 here
        for which in (0, 1):
            got = [r[which][j] for r in res]
            assert all((g[0], g[1]) == (itn, istop) for g in got), (j, which, [(g[0], g[1]) for g in got], (itn, istop))
            w = np.concatenate([g[2] for g in got])
            assert np.all(np.isfinite(w))
            assert np.linalg.norm(w - w_ref) <= max(1e-9, 10 * rtol) * np.linalg.norm(w_ref)


@pytest.mark.parametrize("P", [3, 5])
def test_feast_contour_replicas(hip, P):
    """FEAST with the 4 contour points dealt to P replicas (2+1+1, and 1+1+1+1+0: one rank idle), whole
    operator on every rank, one all-reduce per filtered vector: same eigenvalues as the serial device
    run and as the reference's own run (golden file)."""
    import warnings
    from eigensolvers_amd.distributed import ContourReplicas
    g = load_golden("feast_n100.npz")
    opts = {"linearSystemArgs": {"linearSolver": "gcrotmk", "linearIter": 1000, "linear_tol": 1e-2}}

    def run(ctx, comm):
        A = hip.HipCsrOperator.from_dense(g["A"], ctx=ctx)
        Y = [hip.HipVector(g["guess"][:, i].copy(), dict(opts), ctx=ctx) for i in range(6)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev, Yf, st = hip.feastDiagonalization(A, Y, 8, "legendre", 160.0, 166.0, 1e-10, 20, writeOut=False,
                                                  contourComm=comm)
        return ev, st["outerIter"], len(Yf), Yf[0].array

    ev_s, it_s, n_s, y_s = run(hip.HipContext.default(), None)
    grp = LoopbackGroup(P)
    try:
        res = grp.run(lambda rank, ctx: run(ctx, ContourReplicas(ctx)))
    finally:
        grp.close()
    for ev, it, n, y0 in res:
        np.testing.assert_array_equal(ev, res[0][0])
        np.testing.assert_array_equal(y0, res[0][3])
        # the sum over contour points is grouped by rank: rounding-level changes, amplified outside the
        # window where FEAST does not converge the values (the inner solves stop at rtol 1e-2)
        inside = (ev_s >= 160.0) & (ev_s <= 166.0)
        assert inside.sum() == 3
        np.testing.assert_allclose(ev[inside], ev_s[inside], rtol=1e-9)
        np.testing.assert_allclose(ev[inside], g["ev"][inside], rtol=1e-9)
        np.testing.assert_allclose(ev, ev_s, rtol=1e-6)
        assert it == it_s == int(g["outerIter"]) and n == n_s == int(g["nvec"])


def test_eight_ranks_wide_operator(hip):
    """Eight ranks of an operator wide enough (4.2e6 columns) that every rank's slab takes the
    column-split path on its own (30 row blocks for 256 CUs), with the all-gather overlap, ragged
    slabs and the padded stride - the shape of the 8-GPU benchmark run, on one GPU."""
    N, P = 4_200_003, 8
    single = hip.HipCsrOperator.generate(N, 16, seed=11)
    single.set_variant(2)
    x = np.random.default_rng(6).standard_normal(N)
    y_ref = hip.HipVector(x).applyOp(single).array
    opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 60, "linear_tol": 1e-30}}
    b_full = guess_vector(N, 3) / np.linalg.norm(guess_vector(N, 3))
    grp = LoopbackGroup(P)

    def body(rank, ctx):
        b, e = row_range(N, P, rank)
        H = hip.HipCsrOperator.generate(N, 16, seed=11, row_begin=b, row_end=e, ctx=ctx)
        y = hip.HipVector(x[b:e], ctx=ctx).applyOp(H).array
        X = hip.HipVector(x[b:e], ctx=ctx)
        # a fixed number of MINRES iterations (the tolerance cannot be met): exercises KA with slabs + combine
        try:
            hip.HipVector.solve(H, hip.HipVector(b_full[b:e].copy(), dict(opts), ctx=ctx), 0.02)
            its = -1
        except UserWarning:
            its = 60
        return float(np.max(np.abs(y - y_ref[b:e])) / np.max(np.abs(y_ref))), H.last_variant(), X.vdot(X), its

    try:
        res = grp.run(body)
    finally:
        grp.close()
    for err, variant, dot, its in res:
        assert variant == "column-window-blocked(workgroup)"
        assert err < 1e-14
        assert dot == res[0][2] and abs(dot - np.dot(x, x)) < 1e-10 * np.dot(x, x)
        assert its == 60


def test_block_product_and_block_solve_on_two_ranks(hip, monkeypatch):
    """The block path on a row partition: the interleaved operand block is all-gathered (one collective
    for all 8 columns), both block kernels run on remapped columns, and the lock-step MINRES takes the
    same decisions on both ranks and returns the single-GPU solutions."""
    N, P, k = 120_000, 2, 5
    single = hip.HipCsrOperator.generate(N, 32, seed=5)
    rng = np.random.default_rng(8)
    Xh = rng.standard_normal((N, k))
    Bh = Xh / np.linalg.norm(Xh, axis=0)
    ref = [hip.HipVector(Xh[:, j].copy()).applyOp(single).array for j in range(k)]
    opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-8}}
    sols = hip.HipVector.solveBlock(single, [hip.HipVector(Bh[:, j].copy(), dict(opts)) for j in range(k)], 0.02)
    its_ref = [w.last_solve_stats["iterations"] for w in sols]
    sols = [w.array for w in sols]
    grp = LoopbackGroup(P)

    def body(rank, ctx):
        b, e = row_range(N, P, rank)
        H = hip.HipCsrOperator.generate(N, 32, seed=5, row_begin=b, row_end=e, ctx=ctx)
        out = {}
        for bv in (1, 2):
            H.set_block_variant(bv)
            Y = H.apply_block([hip.HipVector(Xh[b:e, j].copy(), ctx=ctx)._buf for j in range(k)])
            out[f"y{bv}"] = [hip.HipVector(y).array for y in Y]
            W = hip.HipVector.solveBlock(H, [hip.HipVector(Bh[b:e, j].copy(), dict(opts), ctx=ctx) for j in range(k)], 0.02)
            out[f"w{bv}"] = [w.array for w in W]
            out[f"it{bv}"] = [w.last_solve_stats["iterations"] for w in W]
            out[f"coll{bv}"] = W[0].last_solve_stats["collectives"]
            out[f"kind{bv}"] = H.block_info()["variant"]
        return out, (b, e)

    try:
        res = grp.run(body)
    finally:
        grp.close()
    for bv, kind in ((1, "row-owner"), (2, "column-window-blocked")):
        assert all(o[f"kind{bv}"] == kind for o, _ in res)
        assert res[0][0][f"it{bv}"] == res[1][0][f"it{bv}"]
        # the block solve: per iteration one exchange of the interleaved operand block (with the ranks' shares of the K
        # <y_j,y_j>) and ONE all-reduce of two records (<v,y>, lagged <x,x>) for all columns; whole chunks of 16
        done = 16 * -(-(max(res[0][0][f"it{bv}"]) + 1) // 16)
        assert res[0][0][f"coll{bv}"] == 2 * done, (res[0][0][f"coll{bv}"], done)
        for j in range(k):
            y = np.concatenate([o[f"y{bv}"][j] for o, _ in res])
            assert np.max(np.abs(y - ref[j])) <= 1e-13 * np.max(np.abs(ref[j]))
            w = np.concatenate([o[f"w{bv}"][j] for o, _ in res])
            assert np.linalg.norm(w - sols[j]) <= 1e-6 * np.linalg.norm(sols[j])
            assert abs(res[0][0][f"it{bv}"][j] - its_ref[j]) <= 2


def test_config4_operator_on_eight_ranks(hip):
    """BASELINE config #4 shape on one GPU: the N = 1e7, 64 nnz/row operator row-partitioned over 8
    loopback ranks (each builds its own 1.25e6-row slab on the device), a few products and shifted
    products against the whole operator held by a ninth context, and 20 MINRES iterations with the
    fused reductions.  Size-independent checks only (no oracle at this size)."""
    N, P = 10_000_000, 8
    whole = hip.HipCsrOperator.generate(N, 64, seed=7)
    rng = np.random.default_rng(12)
    x = rng.standard_normal(N)
    y_ref = hip.HipVector(x).applyOp(whole).array
    scale = np.max(np.abs(y_ref))
    nnz_whole = whole.nnz
    del whole
    opts = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 20, "linear_tol": 1e-30}}
    grp = LoopbackGroup(P)

    def body(rank, ctx):
        b, e = row_range(N, P, rank)
        H = hip.HipCsrOperator.generate(N, 64, seed=7, row_begin=b, row_end=e, ctx=ctx)
        X = hip.HipVector(x[b:e], ctx=ctx)
        y = X.applyOp(H)
        err = float(np.max(np.abs(y.array - y_ref[b:e])))
        buf = ctx.alloc(e - b)
        H.apply_shifted(0.02, X._buf, buf)
        errs = float(np.max(np.abs(hip.HipVector(buf).array - (0.02 * x[b:e] - y_ref[b:e]))))
        sym = X.vdot(y)                                    # <x, Hx>, all-reduced
        try:
            hip.HipVector.solve(H, hip.HipVector(x[b:e] / np.linalg.norm(x), dict(opts), ctx=ctx), 0.02)
            its = -1
        except UserWarning:
            its = 20
        return err, errs, sym, its, H.nnz, H.last_variant()

    try:
        res = grp.run(body)
    finally:
        grp.close()
    assert sum(r[4] for r in res) == nnz_whole
    for err, errs, sym, its, nnz, variant in res:
        assert err <= 1e-13 * scale and errs <= 1e-13 * scale
        assert sym == res[0][2] and abs(sym - np.dot(x, y_ref)) <= 1e-10 * abs(np.dot(x, y_ref)) + 1e-6
        assert its == 20 and variant == "column-window-blocked(workgroup)"


def test_device_group_scalars_on_loopback_ranks(hip):
    """bench.py's barrier / max-over-ranks / sum-over-ranks (distributed.DeviceGroup: the library's own all-reduce,
    no torch) with three ranks."""
    from eigensolvers_amd.distributed import DeviceGroup
    grp = LoopbackGroup(3)

    def body(rank, ctx):
        g = DeviceGroup(ctx)
        g.barrier()
        out = (g.allgather_scalar(rank + 0.5), g.allmax(10.0 * rank), g.allsum(rank + 1.0), g.rank, g.world)
        g.barrier()
        return out

    try:
        res = grp.run(body)
    finally:
        grp.close()
    for r, (gathered, mx, sm, rank, world) in enumerate(res):
        assert gathered == [0.5, 1.5, 2.5] and mx == 20.0 and sm == 6.0 and (rank, world) == (r, 3)


def test_exchange_switch_times_one_ranks_sweeps_without_its_peers(hip):
    """hipeig_comm_set_exchange(0) (round 4, bench.py `phases.sweeps_alone_ms`): a product places the rank's own slice and
    skips the exchange, so ONE rank can run its launches while its peers sit idle - no collective, no wait.  With an
    unchanged operand the gathered buffer still holds the peers' parts of the last real exchange, so the product is the
    true one; after a changed operand it is not (the peers' parts are stale), and switching the exchange back on repairs it."""
    N, P = 300_000, 2
    grp = LoopbackGroup(P)
    x = np.random.default_rng(8).standard_normal(N)

    def body(rank, ctx):
        b, e = row_range(N, P, rank)
        H = hip.HipCsrOperator.generate(N, 32, seed=7, row_begin=b, row_end=e, ctx=ctx)
        X = hip.HipVector(x[b:e], ctx=ctx)
        y0 = X.applyOp(H).array                                 # a real exchange
        out = {}
        if rank == 1:                                            # rank 1 alone: rank 0 makes no call at all meanwhile
            ctx.set_exchange(False)
            out["same"] = float(np.max(np.abs(X.applyOp(H).array - y0)))
            out["stale"] = float(np.max(np.abs((X * 2.0).applyOp(H).array - 2.0 * y0)))
            ctx.set_exchange(True)
        y2 = (X * 2.0).applyOp(H).array                          # collective again, both ranks
        out["back"] = float(np.max(np.abs(y2 - 2.0 * y0)))
        out["scale"] = float(np.max(np.abs(y0)))
        return out

    try:
        res = grp.run(body)
    finally:
        grp.close()
    assert res[1]["same"] <= 1e-13 * res[1]["scale"]
    assert res[1]["stale"] > 1e-3 * res[1]["scale"]               # the peers' half of the operand was NOT doubled
    for o in res:
        assert o["back"] <= 1e-13 * o["scale"]

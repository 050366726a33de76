"""Krylov-basis checkpoints, resume and the optional thick restart of the Lanczos driver
(the reference writes per-iteration dumps at inexact_Lanczos.py:384-393
and has no reader).  Runs on CPU with the oracle's ndarray vector as the backend."""
import json
import os
import warnings

import numpy as np
import pytest
import scipy.linalg as la

import eigensolvers_amd as ea
from eigensolvers_amd import checkpoint as ck
from oracle.numpy_vector import RefVector

ea.AbstractVector.register(RefVector)


def _opts(tol=1e-10):
    return {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": tol}}


def _run(H, guess, L=5, maxit=6, **kw):
    return ea.inexactLanczosDiagonalization(H, RefVector(guess.copy(), _opts()), 0.02, L, maxit, 1e-12,
                                            writeOut=False, **kw)


def test_resume_reproduces_the_uninterrupted_run(tmp_path, gapped4000):
    H, guess = gapped4000
    d = str(tmp_path / "ck")
    ev, Y, st = _run(H, guess, checkpointDir=d, checkpointKeep=0)
    files = sorted(os.listdir(d))
    assert files == [f"krylov_{i:06d}.npz" for i in range(1, st["cumIter"] + 1)]
    assert st["isConverged"] and st["cumIter"] > 5             # more than one restart cycle (L - 1 = 4 per cycle)
    # mid-cycle, last iteration of a cycle (the restart itself is redone) and first of the next cycle
    for it in (2, 4, 5):
        ev2, Y2, st2 = _run(H, guess, resumeFrom=os.path.join(d, f"krylov_{it:06d}.npz"))
        np.testing.assert_array_equal(ev2, ev)
        assert (st2["cumIter"], st2["outerIter"], st2["innerIter"]) == (st["cumIter"], st["outerIter"], st["innerIter"])
        assert st2["residual"] == st["residual"] and st2["isConverged"]
        for a, b in zip(Y, Y2):
            np.testing.assert_array_equal(a.array, b.array)
    # a directory means "the newest checkpoint": the converged one, so no iteration is added
    ev3, Y3, st3 = _run(H, guess, resumeFrom=d)
    np.testing.assert_array_equal(ev3, ev)
    assert st3["cumIter"] == st["cumIter"] and st3["isConverged"]
    np.testing.assert_allclose(Y3[0].array, Y[0].array, rtol=0, atol=1e-15)


def test_checkpoint_contents_and_pruning(tmp_path, gapped4000):
    H, guess = gapped4000
    d = str(tmp_path / "ck")
    ev, Y, st = _run(H, guess, maxit=1, checkpointDir=d)          # default: keep the newest two
    assert sorted(os.listdir(d)) == [f"krylov_{i:06d}.npz" for i in (3, 4)]
    assert ck.latest_checkpoint(d) == os.path.join(d, "krylov_000004.npz")
    c = ck.load_checkpoint(ck.latest_checkpoint(d))
    m = c["Y"].shape[0]
    assert c["Y"].shape == (5, 4000) and c["S"].shape == c["Hm"].shape == c["eigencoefficients"].shape == (m, m)
    vecs = ck.restore_vectors(RefVector(guess.copy(), _opts()), c["Y"])
    np.testing.assert_allclose(RefVector.overlapMatrix(vecs), c["S"], atol=1e-14)
    np.testing.assert_allclose(RefVector.matrixRepresentation(H, vecs), c["Hm"], atol=1e-13)
    # the Ritz data of the file reproduce the returned eigenvalues
    np.testing.assert_array_equal(c["eigenvalues"], ev)
    assert c["status"]["cumIter"] == 4 and c["meta"]["sigma"] == 0.02 and c["meta"]["nranks"] == 1
    assert isinstance(c["status"]["ref"][0], np.ndarray)
    # plain arrays and JSON only: loads with pickles disabled
    with np.load(ck.latest_checkpoint(d), allow_pickle=False) as z:
        json.loads(str(z["status"]))
    assert not [f for f in os.listdir(d) if f.endswith(".tmp")]


def test_resume_argument_checks(tmp_path, gapped4000):
    H, guess = gapped4000
    d = str(tmp_path / "ck")
    _run(H, guess, maxit=1, L=3, checkpointDir=d)
    with pytest.raises(FileNotFoundError):
        _run(H, guess, resumeFrom=str(tmp_path))
    with pytest.raises(ValueError, match="length"):
        ea.inexactLanczosDiagonalization(H[:100, :100].tocsr(), RefVector(guess[:100].copy(), _opts()), 0.02, 3, 1, 1e-12,
                                         writeOut=False, resumeFrom=d)
    with pytest.raises(ValueError, match="block size"):
        Q = la.qr(np.random.default_rng(5).standard_normal((4000, 2)), mode="economic")[0]
        ea.inexactLanczosDiagonalization(H, [RefVector(Q[:, i].copy(), _opts()) for i in range(2)], 0.02, 3, 1, 1e-12,
                                         writeOut=False, resumeFrom=d)


def test_partitioned_runs_write_one_file_per_rank(tmp_path):
    class Ctx:
        rank, nranks = 1, 4

    class Vec(RefVector):
        ctx = Ctx()

    v = [Vec(np.eye(6)[i], _opts()) for i in range(2)]
    st = {"cumIter": 7, "nBlock": 1, "ref": [np.array([1.0])], "residual": np.inf}
    name = ck.save_checkpoint(str(tmp_path), v, np.eye(2), np.eye(2), np.eye(2), np.ones(2), st)
    assert os.path.basename(name) == "krylov_000007.r1of4.npz"
    assert ck.latest_checkpoint(str(tmp_path), 1, 4) is None             # ranks 0, 2, 3 have not written iteration 7 yet
    for r in (0, 2, 3):
        Ctx.rank = r
        ck.save_checkpoint(str(tmp_path), v, np.eye(2), np.eye(2), np.eye(2), np.ones(2), st)
    assert ck.latest_checkpoint(str(tmp_path), 1, 4) == name
    c = ck.load_checkpoint(name)
    assert c["status"]["residual"] == np.inf and c["meta"]["rank"] == 1


def test_thick_restart(gapped4000):
    H, guess = gapped4000
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # accurate solves: both restarts converge to the same eigenvalue in about as many iterations
        ev_s, Y_s, st_s = _run(H, guess, L=4, maxit=12)
        ev_t, Y_t, st_t = _run(H, guess, L=4, maxit=12, thickRestart=2)
        assert st_s["isConverged"] and st_t["isConverged"]
        assert abs(ev_t[0] - ev_s[0]) <= 1e-10 * abs(ev_s[0])
        assert abs(st_t["cumIter"] - st_s["cumIter"]) <= 2 and len(Y_t) >= len(Y_s)
        assert ea.true_residual_norms(H, ev_t, Y_t, 1)[0] < 1e-6
        # sloppy solves (rtol 1e-3) and a short cycle: discarding the basis at every restart stalls,
        # keeping two more Ritz vectors converges
        loose = {"linearSystemArgs": {"linearSolver": "minres", "linearIter": 2000, "linear_tol": 1e-3}}
        run = lambda k: ea.inexactLanczosDiagonalization(H, RefVector(guess.copy(), dict(loose)), 0.02, 4, 15, 1e-12,
                                                         writeOut=False, thickRestart=k)
        _, _, st_simple = run(0)
        ev_thick, _, st_thick = run(2)
    assert not st_simple["isConverged"] and st_simple["cumIter"] == 45
    assert st_thick["isConverged"] and st_thick["cumIter"] < 45
    assert abs(ev_thick[0] - ev_s[0]) < 1e-5


def test_partitioned_resume_uses_the_newest_iteration_all_ranks_have(tmp_path):
    """ranks write and prune their own files; after a crash between two ranks' writes the
    newest files differ per rank.  Every rank must resume from the same (newest complete) iteration."""
    d = str(tmp_path)
    for it, ranks in ((7, (0, 1, 2)), (8, (0, 1, 2)), (9, (0, 2))):          # rank 1 died before writing 9
        for r in ranks:
            open(ck.checkpoint_name(d, it, r, 3), "wb").close()
    open(os.path.join(d, "krylov_000011.npz"), "wb").close()                 # an unpartitioned run's file: ignored
    for r in range(3):
        assert ck.latest_checkpoint(d, r, 3) == ck.checkpoint_name(d, 8, r, 3)
    assert ck.latest_checkpoint(d, 0, 1) == os.path.join(d, "krylov_000011.npz")
    os.remove(ck.checkpoint_name(d, 8, 2, 3))
    assert ck.latest_checkpoint(d, 1, 3) == ck.checkpoint_name(d, 7, 1, 3)
    assert ck.latest_checkpoint(d, 0, 2) is None


def test_resume_refuses_other_run_parameters(tmp_path, gapped4000):
    H, guess = gapped4000
    d = str(tmp_path / "ck")
    _run(H, guess, maxit=1, checkpointDir=d)
    with pytest.raises(ValueError, match="sigma"):
        ea.inexactLanczosDiagonalization(H, RefVector(guess.copy(), _opts()), 0.03, 5, 6, 1e-12, writeOut=False, resumeFrom=d)
    with pytest.raises(ValueError, match="L="):
        ea.inexactLanczosDiagonalization(H, RefVector(guess.copy(), _opts()), 0.02, 7, 6, 1e-12, writeOut=False, resumeFrom=d)
    with pytest.raises(ValueError, match="eConv"):
        ea.inexactLanczosDiagonalization(H, RefVector(guess.copy(), _opts()), 0.02, 5, 6, 1e-9, writeOut=False, resumeFrom=d)

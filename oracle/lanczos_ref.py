"""Oracle (test infrastructure): CPU restatement of the reference's outer Lanczos loop.

Follows ``inexactLanczosDiagonalization`` (inexact_Lanczos.py:229-443) and the host
helpers it calls (util_funcs.py:208-231, 233-247, 249-289, 305-358, 360-385), written
backend-agnostically exactly like the reference: the vector backend is
``type(v0[0])`` and only its static hooks are used.  With ``oracle.numpy_vector.RefVector``
this IS the CPU comparator of the product; it is pinned against the real reference by
``tests/golden/*`` (see ``tests/test_oracle_golden.py``).

Deliberate, documented differences from the reference HEAD:
* no file output and no per-iteration TTNS checkpoint (``writeOut=False``,
  ``saveTNSsEachIteration=False`` semantics; inexact_Lanczos.py:303-306, 384-393);
* the GS-lindep exit does not ``del`` unbound locals (inexact_Lanczos.py:358) - it just
  returns NaN eigenvalues for the current basis as line 357 does.

Never imported by ``eigensolvers_amd``.
"""
import time
import warnings

import numpy as np
import scipy.linalg as sla

LINDEP = 1e-14


# ---- util_funcs.py restatements ---------------------------------------------------
def basis_transformation(bases, coeffs):                       # util_funcs.py:208-231
    cls = type(bases[0])
    coeffs = np.asarray(coeffs)
    if coeffs.ndim == 1:
        if len(coeffs) == 1 and coeffs[0] == 1.0:
            return [bases]                                     # quirk: list inside a list
        return [cls.linearCombination(bases, coeffs)]
    return [cls.linearCombination(bases, coeffs[:, j]) for j in range(coeffs.shape[1])]


def lowdin_ortho(S, tol=LINDEP):                               # util_funcs.py:233-247
    lam, U = sla.eigh(S)
    keep = lam > tol
    return keep, bool(np.all(keep)), U[:, keep] * lam[keep] ** (-0.5)


def lowdin_ortho_matrix(S, status):                            # util_funcs.py:346-358
    _, indep, uS = lowdin_ortho(S)
    status["lindep"] = not indep
    return status, uS


def diagonalize_hamiltonian(X, Hmat):                          # util_funcs.py:360-385
    return sla.eigh(X.T.conj() @ Hmat @ X)


def eigenvalue_residual(ev, reference):                        # util_funcs.py:249-289
    num = 0.0
    den = 0.0
    for i in range(len(ev)):
        num += abs(reference[i] - ev[i])
        den += abs(ev[i])
    return num / den


def pick_close_to_sigma(sigma):                                # util_funcs.py:330-344
    return lambda T, vectors, ev: np.argsort(np.abs(ev - sigma))


def pick_max_overlap(reference_vector):                        # util_funcs.py:305-328
    def pick(T, vectors, ev):
        ov = np.zeros(T.shape[0], dtype=T[0].dtype)
        for i in range(T.shape[0]):
            ov[i] = vectors[i].vdot(reference_vector)
        return np.argsort(-abs(T.T.conj() @ ov))
    return pick


# ---- inexact_Lanczos.py restatements ------------------------------------------------
def init_status(status, guess, nBlock):                        # inexact_Lanczos.py:23-82
    st = {"ref": [], "residual": np.inf, "nBlock": nBlock,
          "flagAddition": guess.hasExactAddition,
          "outerIter": 0, "innerIter": 0, "cumIter": 0, "iBlock": 0,
          "zeroVector": False, "isConverged": False, "lindep": False,
          "futileRestarts": 0, "startTime": time.time(), "runTime": 0.0,
          "KSmaxD": [], "fitmaxD": None, "phase": 1}
    if status is not None:
        st.update(status)
    return st


def generate_subspace(Hop, vec, sigma, eConv):                 # inexact_Lanczos.py:84-105
    cls = type(vec)
    out = cls.solve(Hop, vec, sigma)
    if cls.norm(out) > 0.001 * eConv:
        return cls.normalize(out), True
    return out, False


def check_convergence(ev, eConv, status):                      # inexact_Lanczos.py:115-143
    nB = status["nBlock"]
    block = np.sort(ev[0:nB])
    converged = False
    if status["cumIter"] > 1:
        res = eigenvalue_residual(block, status["ref"][-1])
        status["residual"] = res
        converged = res <= eConv
    status["isConverged"] = converged
    status["runTime"] = time.time() - status["startTime"]
    status["ref"].append(block)
    if len(status["ref"]) > 2:
        status["ref"].pop(0)
    return status


def terminate_restart(block_energies, eConv, status, num=3):   # inexact_Lanczos.py:167-194
    if status["lindep"]:
        if eigenvalue_residual(block_energies, status["ref"][0]) > max(1e-9, eConv):
            status["futileRestarts"] += 1
    if status["futileRestarts"] > num:
        warnings.warn("Lindep and did not have fruitful restarts")
        return True
    return False


def analyze_status(status, maxit, L):                          # inexact_Lanczos.py:197-222
    if status["isConverged"]:
        return False
    if status["outerIter"] == maxit - 1 and status["innerIter"] == L - 1:
        print("Alert: Lanczos iterations is not converged!")
        return False
    return True


def inexact_lanczos(H, v0, sigma, L, maxit, eConv, checkFitTol=1e-7, Hsolve=None,
                    pick=None, status=None, history=None):
    """inexact_Lanczos.py:229-443.  Returns ``(ev, Ylist, status)``.

    ``history`` (optional list) receives ``(cumIter, ev.copy())`` after every
    diagonalisation - used only to write/check golden traces.
    """
    if not isinstance(v0, (list, tuple, np.ndarray)):
        v0 = [v0]
    if Hsolve is None:
        Hsolve = H
    cls = type(v0[0])
    nBlock = len(v0)

    Y = list(v0)                                               # shallow copy, :287
    S = cls.overlapMatrix(Y)
    if not np.allclose(S, np.eye(nBlock), rtol=1e-3, atol=1e-3):
        if nBlock > 1:
            raise RuntimeError(f"Input vectors not orthogonalized: Smat={S}")
        Y[0].normalize()                                       # mutates the caller's guess, :294
        S[0, 0] = 1
    Hm = cls.matrixRepresentation(H, Y)
    status = init_status(status, Y[0], nBlock)
    if pick is None:
        pick = pick_close_to_sigma(sigma)

    ev = None
    T = None
    lindep_problem = False
    go_on = True
    for outer in range(maxit):
        status["outerIter"] = outer
        status["KSmaxD"] = [Y[0].maxD]
        status["fitmaxD"] = None
        nonzero = True
        for inner in range(1, L):
            status["innerIter"] = inner
            status["cumIter"] += 1
            new = []
            for ib in range(1, nBlock + 1):                    # :319-327
                out, nonzero = generate_subspace(Hsolve, Y[-ib], sigma, eConv)
                if not nonzero:
                    status["zeroVector"] = True
                    warnings.warn(f"Alert: zero vector: ||inv(H-sigma)vec||={cls.norm(out):5.3e}")
                    break
                new.append(out)
            if not nonzero:
                break
            lindep_problem = False
            for ib in range(nBlock):                           # :335-350
                status["iBlock"] = ib
                q = cls.orthogonalize_against_set(new[ib], Y)
                if q is None:
                    lindep_problem = True
                    break
                Y.append(q.compress())
                status["KSmaxD"].append(Y[-1].maxD)
                S = cls.extendOverlapMatrix(Y, S)
                Hm = cls.extendMatrixRepresentation(H, Y, Hm)
            if lindep_problem:                                 # :356-359
                ev = np.array([np.nan] * len(Y))
                break
            status, uS = lowdin_ortho_matrix(S, status)        # :367-370
            assert not status["lindep"]
            ev, uv = diagonalize_hamiltonian(uS, Hm)
            T = uS @ uv
            idx = pick(T, Y, ev)                               # :373-376
            assert len(idx) == len(ev)
            ev = ev[idx]
            T = T[:, idx]
            status = check_convergence(ev, eConv, status)      # :380-381
            if history is not None:
                history.append((status["cumIter"], ev.copy()))
            go_on = analyze_status(status, maxit, L)
            if not go_on:
                break
        if lindep_problem:
            break
        if not go_on:                                          # :400-412
            Y = basis_transformation(Y, T)
            S = cls.overlapMatrix(Y)
            if not np.allclose(S, np.eye(len(Y)), rtol=checkFitTol, atol=checkFitTol):
                warnings.warn(f"Alert:Final eigenvectors are not properly fitted. S=\n{S}")
            status["fitmaxD"] = [v.maxD for v in Y]
            break
        # simple restart from the nBlock picked Ritz vectors, :414-436
        guesses = []
        for ib in range(nBlock):
            g = basis_transformation(Y, T[:, ib])
            guesses.append(cls.normalize(g[0]))
        Y = guesses
        S = cls.overlapMatrix(Y)
        Hm = cls.matrixRepresentation(H, Y)
        if not np.allclose(S, np.eye(len(Y)), rtol=checkFitTol, atol=checkFitTol):
            warnings.warn(f"Alert:Final eigenvectors are not properly fitted. S=\n{S}")
            break
        evNew = sla.eigvalsh(Hm, S)
        if terminate_restart(evNew, eConv, status):
            break
        status["fitmaxD"] = [v.maxD for v in Y]
    return ev, Y, status

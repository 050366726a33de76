"""Krylov-basis checkpoints for the restarted Lanczos driver.

The reference dumps every Krylov vector after every iteration through the TTNS backend's HDF5
writer (inexact_Lanczos.py:384-393, ``tns_{nCum}_{ivector}.h5`` with status, eigencoefficients and
eigenvalues attached) and has no reader.  This is the array-backend counterpart plus the missing
reader: one ``.npz`` per cumulative iteration holding

* ``Y``      (m, n_local) - the basis vectors (the rank's row slice on a partitioned run),
* ``S``, ``Hm`` (m, m)    - overlap matrix and projected operator of that basis,
* ``eigencoefficients`` (m, m), ``eigenvalues`` (m,) - the Ritz data of that iteration,
* ``status``              - the status dictionary as JSON (arrays as lists),
* ``meta``                - JSON: sigma, L, eConv, nBlock, partition (rank, nranks), format version.

Resuming reproduces the uninterrupted run bit for bit where the backend's arithmetic is reproducible
(ndarray backends; HipVector with operator kernel variants 1-3 - the default variant 4 for large
operators sums a row with LDS atomics and agrees to rounding only).

Files are written to a temporary name and renamed, so a run killed while writing leaves the previous
checkpoint intact; by default only the newest ``keep`` checkpoints are retained.  Nothing here
touches the device: vectors are read through ``.array`` and rebuilt through the backend's
constructor ``type(template)(array, template.options)`` - the signature every array backend of
the reference has (numpyVector.py:25) - or its ``fromArray`` hook when it defines one.
"""
import glob
import json
import os
import re
import time

import numpy as np

FORMAT_VERSION = 1

__all__ = ["save_checkpoint", "load_checkpoint", "latest_checkpoint", "restore_vectors",
           "checkpoint_name", "check_meta"]


def _partition_of(vec):
    ctx = getattr(vec, "ctx", None)
    return int(getattr(ctx, "rank", 0)), int(getattr(ctx, "nranks", 1))


def checkpoint_name(saveDir, cumIter, rank=0, nranks=1):
    tag = "" if nranks == 1 else f".r{rank}of{nranks}"
    return os.path.join(saveDir, f"krylov_{int(cumIter):06d}{tag}.npz")


def _jsonable(value):
    if isinstance(value, np.ndarray):
        return value.tolist()
    if isinstance(value, (np.floating, np.integer, np.bool_)):
        return value.item()
    if isinstance(value, (list, tuple)):
        return [_jsonable(v) for v in value]
    if isinstance(value, dict):
        return {str(k): _jsonable(v) for k, v in value.items()}
    if isinstance(value, float) and not np.isfinite(value):
        return repr(value)                       # inf / nan survive as strings
    return value


def _status_from_json(text):
    st = json.loads(text)
    for key in ("residual", "runTime", "startTime"):
        if isinstance(st.get(key), str):
            st[key] = float(st[key])
    st["ref"] = [np.asarray(r, dtype=float) for r in st.get("ref", [])]
    return st


def save_checkpoint(saveDir, Y, S, Hm, coeffs, ev, status, sigma=None, L=None, eConv=None, keep=2):
    """Write the checkpoint of the current iteration; returns the file name."""
    os.makedirs(saveDir, exist_ok=True)
    rank, nranks = _partition_of(Y[0])
    name = checkpoint_name(saveDir, status["cumIter"], rank, nranks)
    basis = np.stack([np.asarray(v.array) for v in Y])
    meta = {"version": FORMAT_VERSION, "sigma": sigma, "L": L, "eConv": eConv,
            "nBlock": status.get("nBlock"), "rank": rank, "nranks": nranks,
            "backend": type(Y[0]).__name__, "written": time.time()}
    tmp = name + ".tmp"
    with open(tmp, "wb") as fh:
        np.savez(fh, Y=basis, S=np.asarray(S), Hm=np.asarray(Hm), eigencoefficients=np.asarray(coeffs),
                 eigenvalues=np.asarray(ev), status=np.array(json.dumps(_jsonable(status))),
                 meta=np.array(json.dumps(_jsonable(meta))))
        fh.flush()
        os.fsync(fh.fileno())
    os.replace(tmp, name)
    if keep:
        tag = "" if nranks == 1 else f".r{rank}of{nranks}"
        mine = sorted(glob.glob(os.path.join(saveDir, f"krylov_??????{tag}.npz")))
        if nranks == 1:
            doomed = mine[:-keep]
        else:
            # Ranks write and prune independently.  A rank that pruned by its OWN newest files could delete the last
            # iteration every rank still has (keep = 1, or a crash between two ranks' writes): the intersection
            # latest_checkpoint() takes would then be empty although a complete state existed a moment earlier.  So on a
            # partitioned run only iterations older than the newest COMPLETE one (minus keep - 1) are removed, and at
            # least two generations are kept.
            keep = max(int(keep), 2)
            common = _iterations_of(saveDir, 0, nranks)
            for r in range(1, nranks):
                common &= _iterations_of(saveDir, r, nranks)
            floor = sorted(common)[-keep] if len(common) >= keep else None
            doomed = [] if floor is None else [f for f in mine if int(os.path.basename(f)[7:13]) < floor]
        for old in doomed:
            os.remove(old)
    return name


def load_checkpoint(path):
    """Read a checkpoint written by :func:`save_checkpoint` (no pickles are involved)."""
    with np.load(path, allow_pickle=False) as z:
        out = {k: z[k] for k in ("Y", "S", "Hm", "eigencoefficients", "eigenvalues")}
        out["status"] = _status_from_json(str(z["status"]))
        out["meta"] = json.loads(str(z["meta"]))
    if out["meta"].get("version") != FORMAT_VERSION:
        raise ValueError(f"{path}: checkpoint format {out['meta'].get('version')} is not {FORMAT_VERSION}")
    return out


def _iterations_of(saveDir, rank, nranks):
    tag = "" if nranks == 1 else f".r{rank}of{nranks}"
    its = set()
    for f in glob.glob(os.path.join(saveDir, f"krylov_*{tag}.npz")):
        m = re.fullmatch(rf"krylov_(\d+){re.escape(tag)}\.npz", os.path.basename(f))
        if m:
            its.add(int(m.group(1)))
    return its


def latest_checkpoint(saveDir, rank=0, nranks=1):
    """This rank's file of the newest iteration that is COMPLETE, i.e. for which every rank of the
    partition has a file (None when there is none).  Ranks write and prune independently, so after a
    crash between two ranks' writes their newest files can belong to different iterations; resuming each
    rank from its own newest file would make the ranks issue different numbers of collectives.  All
    ranks see the same directory and therefore pick the same iteration."""
    common = _iterations_of(saveDir, 0, nranks)
    for r in range(1, nranks):
        common &= _iterations_of(saveDir, r, nranks)
    if not common:
        return None
    return checkpoint_name(saveDir, max(common), rank, nranks)


def check_meta(meta, sigma, L, eConv, nranks, path="checkpoint"):
    """Refuse to continue a run whose parameters differ from the ones the checkpoint was written with.  Floats are compared
    exactly (they survive the JSON round trip bit for bit); a key the checkpoint does not carry is not checked.
    Remaining failure mode of a partitioned resume: ranks whose directories are not the same shared directory see
    different file sets and may pick different iterations - resume from a directory every rank can read."""
    for key, now in (("sigma", sigma), ("L", L), ("eConv", eConv), ("nranks", nranks)):
        was = meta.get(key)
        if was is not None and now is not None and was != now:
            raise ValueError(f"{path}: written with {key}={was!r}, resumed with {key}={now!r}")


def restore_vectors(template, basis):
    """Backend vectors like ``template`` from the rows of ``basis``."""
    cls = type(template)
    make = getattr(cls, "fromArray", None)
    if make is not None:
        return [make(template, np.array(row)) for row in basis]
    return [cls(np.array(row), template.options) for row in basis]

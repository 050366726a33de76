"""Synthetic inputs for the hot path: seeded "gapped random-sparse Hermitian" operators.

The reference ships no sparse test matrices (its tests use dense ``Q^T diag(ev) Q``,
unittests/test_lanczos.py:14-20).  The north-star configurations need random-sparse
Hermitian CSR operators with a spectral gap around the shift, at
sizes (N = 1e7, 6.5e8 non-zeros) that cannot be assembled through scipy on the host, so
the generator is defined by integer hashing and exists twice, bit for bit identical:

* ``hipeig_csr_generate`` (eigensolvers_amd/csrc/generate.hip) builds any row range on the
  device - this is what benchmarks and multi-GPU runs use;
* ``gapped_csr_host`` below builds the same matrix with NumPy for the CPU comparator.

Structure: K pseudo-random permutations pi_k of [0,N) (4-round Feistel networks with
cycle walking).  Row i holds the forward edge (i, pi_k(i)) when keep(k,i) and the inverse
edge (i, pi_k^-1(i)) when keep(k, pi_k^-1(i)); both ends of an edge see the same value, so
H = H^T by construction, with ~nnz_row off-diagonals per row (binomially distributed row
lengths) in uniformly random columns.  Values have unit variance scaled to
eps/sqrt(nnz_row); the diagonal is +-(1..10) except ``ntargets`` rows that hold
``linspace(-0.2, 0.2)`` - the mid-spectrum cluster the shift sigma = 0.02 sits in.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = 0x9E3779B97F4A7C15


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _mix64_int(z):
    z &= 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def gapped_params(N, nnz_row=64, seed=7, eps=0.05, ntargets=16, K=None, thresh24=None):
    """Parameters shared by the device and the host generator."""
    if K is None:
        K = int(np.ceil(nnz_row / 2 * 9 / 8))            # keep probability ~ 8/9
    if thresh24 is None:
        thresh24 = int(round(nnz_row / (2.0 * K) * (1 << 24)))
    thresh24 = min(thresh24, 1 << 24)
    targets = np.linspace(-0.2, 0.2, ntargets) if ntargets else np.zeros(0)
    return {"N": int(N), "K": int(K), "seed": int(seed) & 0xFFFFFFFFFFFFFFFF, "eps": float(eps),
            "thresh24": int(thresh24), "targets": targets}


def _keys(p):
    st = p["seed"]
    out = []
    for _ in range(6 * p["K"] + 2):
        st = (st + _GOLD) & 0xFFFFFFFFFFFFFFFF
        out.append(_mix64_int(st))
    K = p["K"]
    rk = np.array(out[:4 * K], dtype=np.uint64).reshape(K, 4)
    keep = np.array(out[4 * K:5 * K], dtype=np.uint64)
    val = np.array(out[5 * K:6 * K], dtype=np.uint64)
    return rk, keep, val, np.uint64(out[6 * K]), np.uint64(out[6 * K + 1])


def _feistel(v, rk4, hb, N, inverse):
    mask = np.uint64((1 << hb) - 1)
    hbu = np.uint64(hb)
    v = v.copy()
    todo = np.ones(v.shape, dtype=bool)
    while todo.any():
        cur = v[todo]
        L, R = cur >> hbu, cur & mask
        if not inverse:
            for r in range(4):
                L, R = R, L ^ (_mix64(R ^ rk4[r]) & mask)
        else:
            for r in (3, 2, 1, 0):
                L, R = R ^ (_mix64(L ^ rk4[r]) & mask), L
        cur = (L << hbu) | R
        v[todo] = cur
        todo[todo] = cur >= np.uint64(N)
    return v


def gapped_csr_host(N, nnz_row=64, seed=7, row_begin=0, row_end=None, **kw):
    """Rows [row_begin,row_end) of the synthetic operator as a scipy CSR matrix (N columns).

    Duplicate (row, col) entries are kept as separate stored elements, exactly like the
    device layout (``H @ x`` sums them)."""
    import scipy.sparse as sp
    p = gapped_params(N, nnz_row, seed, **kw)
    K, thr = p["K"], np.uint64(p["thresh24"])
    rk, keepk, valk, diagkey, signkey = _keys(p)
    row_end = N if row_end is None else row_end
    rows = np.arange(row_begin, row_end, dtype=np.uint64)
    bits = 1
    while (1 << bits) < N:
        bits += 1
    hb = (bits + 1) // 2
    q = p["thresh24"] / 16777216.0
    vscale = p["eps"] * 1.7320508075688772 / (65535.0 * np.sqrt(2.0 * K * q))

    def keep(k, s):
        return (_mix64(s ^ keepk[k]) >> np.uint64(40)) < thr

    def value(k, s):
        h = _mix64(s ^ valkey_of[k])
        f = np.uint64(0xFFFF)
        tot = ((h & f) + ((h >> np.uint64(16)) & f) + ((h >> np.uint64(32)) & f) + (h >> np.uint64(48)))
        return (tot.astype(np.int64) - 131070).astype(np.float64) * vscale

    valkey_of = valk
    R, Cc, Sl, V = [], [], [], []
    for k in range(K):
        fk = keep(k, rows)
        src = rows[fk]
        R.append(src)
        Cc.append(_feistel(src, rk[k], hb, N, inverse=False))
        Sl.append(np.full(src.shape, 2 * k, dtype=np.uint64))
        V.append(value(k, src))
        inv = _feistel(rows, rk[k], hb, N, inverse=True)
        ik = keep(k, inv)
        R.append(rows[ik])
        Cc.append(inv[ik])
        Sl.append(np.full(int(ik.sum()), 2 * k + 1, dtype=np.uint64))
        V.append(value(k, inv[ik]))
    # diagonal
    nt = len(p["targets"])
    stride = max(N // nt, 1) if nt else 1
    first = stride // 2
    u = (_mix64(rows ^ diagkey) >> np.uint64(11)).astype(np.float64) * 1.1102230246251565e-16
    mag = 1.0 + 9.0 * u
    d = np.where((_mix64(rows ^ signkey) & np.uint64(1)) == 1, -mag, mag)
    if nt:
        ri = rows.astype(np.int64)
        rel = ri - first
        is_t = (rel >= 0) & (rel % stride == 0) & (rel // stride < nt)
        d[is_t] = p["targets"][(rel[is_t] // stride)]
    R.append(rows)
    Cc.append(rows)
    Sl.append(np.full(rows.shape, 2 * K, dtype=np.uint64))
    V.append(d)
    R = np.concatenate(R).astype(np.int64) - row_begin
    Cc = np.concatenate(Cc)
    Sl = np.concatenate(Sl)
    V = np.concatenate(V)
    order = np.lexsort((Sl, Cc, R))
    R, Cc, V = R[order], Cc[order], V[order]
    counts = np.bincount(R, minlength=row_end - row_begin)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    M = sp.csr_matrix((V, Cc.astype(np.int32), indptr), shape=(row_end - row_begin, N))
    return M


def guess_vector(N, seed=1, row_begin=0, row_end=None):
    """Seeded starting vector (default_rng(1).standard_normal(N)), sliced."""
    row_end = N if row_end is None else row_end
    return np.random.default_rng(seed).standard_normal(N)[row_begin:row_end]


def dense_test_matrix(n=100, seed=1212, ev=None):
    """The reference's dense test operator A = Q^T diag(ev) Q (unittests/test_lanczos.py:15-20)."""
    import scipy.linalg as la
    if ev is None:
        ev = np.linspace(1, 200, n)
    np.random.seed(seed)
    Q = la.qr(np.random.rand(n, n))[0]
    return Q.T @ np.diag(ev) @ Q, ev


def sinc_dvr_harmonic(N=45, xrange=(-10.0, 10.0)):
    """H = -d^2/dx^2 + x^2 in the Colbert-Miller sinc-DVR on N equidistant points.

    Stands in for the reference's in-house ``basis.SincInfInf`` (absent from the reference tree;
    unittests/test_stateFollowingHO.py:14-21): T_ij = (-1)^(i-j)/dx^2 * {pi^2/3 (i == j),
    2/(i-j)^2 (i != j)} (J. Chem. Phys. 96, 1982 (1992), eq. A7), V = diag(x_i^2).  The exact
    spectrum is 2n + 1.  Parity with the reference's own grid convention is unpinned (the
    module is not available); the spectrum is what the test relies on."""
    x = np.linspace(xrange[0], xrange[1], N)
    dx = x[1] - x[0]
    i = np.arange(N)
    d = i[:, None] - i[None, :]
    with np.errstate(divide="ignore"):
        T = np.where(d == 0, np.pi ** 2 / 3.0, 2.0 / (d.astype(float) ** 2))
    T = T * (-1.0) ** np.abs(d) / dx ** 2
    return T + np.diag(x ** 2), x

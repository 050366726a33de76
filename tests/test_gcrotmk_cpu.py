"""The GCROT(m,k) host logic (eigensolvers_amd/gcrotmk.py: a generator that yields its operator applications and
orthogonalisation steps) driven on the CPU with a NumPy provider of the vector operations: against
scipy.sparse.linalg.gcrotmk - the routine the reference calls (numpyVector.py:161) - and the lock-step driver of several
right-hand sides (the contour solves of one FEAST contour point, feast.py:198-200) against the one-by-one solves."""
import numpy as np
import pytest
import scipy.sparse.linalg as spl

from eigensolvers_amd.gcrotmk import gcrotmk_device, gcrotmk_device_block
from eigensolvers_amd.generators import gapped_csr_host


class NumpyOps:
    """complex ndarrays in place of (re, im) device buffer pairs; the arithmetic of scipy's BLAS calls"""
    dtype = np.complex128

    def __init__(self, n):
        self.n = n

    def new(self):
        return np.empty(self.n, dtype=complex)

    def zeros(self):
        return np.zeros(self.n, dtype=complex)

    def copy(self, a):
        return a.copy()

    def dot(self, a, b):
        return np.vdot(a, b)

    def nrm2(self, a):
        return float(np.linalg.norm(a))

    def axpy(self, alpha, x, y):
        y += alpha * x

    def scal(self, alpha, x):
        x *= alpha

    def scaled(self, alpha, x):
        return alpha * x

    def arnoldi_step(self, vs, w):
        before = float(np.linalg.norm(w))
        h = np.zeros(len(vs), dtype=complex)
        for j, v in enumerate(vs):
            h[j] = np.vdot(v, w)
            w -= h[j] * v
        after = float(np.linalg.norm(w))
        with np.errstate(divide="ignore", invalid="ignore"):
            alpha = 1.0 / after
        if np.isfinite(alpha):
            w *= alpha
        return before, h, after

    def combine(self, coeffs, vecs):
        out = np.zeros(self.n, dtype=complex)
        for c, v in zip(coeffs, vecs):
            out += c * v
        return out


@pytest.fixture(scope="module")
def system():
    n = 1500
    H = gapped_csr_host(n, 16, seed=3)
    z = 0.02 + 0.05j
    A = lambda v: z * v - H @ v
    rng = np.random.default_rng(0)
    bs = [(rng.standard_normal(n) + 0j) for _ in range(5)]
    bs = [b / np.linalg.norm(b) for b in bs]
    return n, A, bs


def test_generator_driver_tracks_scipy_gcrotmk(system):
    n, A, bs = system
    for b in bs[:2]:
        count, count_s = [0], [0]

        def counted(v):
            count[0] += 1
            return A(v)

        def counted_s(v):
            count_s[0] += 1
            return A(v)

        x, info, stats = gcrotmk_device(None, counted, b, n, rtol=1e-8, atol=1e-12, maxiter=200, complex_pairs=True, ops=NumpyOps(n))
        xs, infos = spl.gcrotmk(spl.LinearOperator((n, n), matvec=counted_s, dtype=complex), b, rtol=1e-8, atol=1e-12, maxiter=200)
        assert info == infos == 0                              # SciPy's convention: 0 = converged
        assert np.linalg.norm(A(x) - b) <= 1e-8
        assert np.linalg.norm(x - xs) <= 1e-7 * np.linalg.norm(xs)
        assert stats["matvecs"] == count[0]
        assert abs(count[0] - count_s[0]) <= max(3, count_s[0] // 50)      # the same algorithm: the same number of products (to rounding)


def test_lock_step_driver_is_the_single_solves_bit_for_bit(system):
    """Same arithmetic per right-hand side whether the products are handed over one by one or collected into blocks: the
    lock-step driver only changes WHEN things run, so solutions, info and product counts are identical."""
    n, A, bs = system
    single = [gcrotmk_device(None, A, b, n, rtol=1e-7, atol=1e-12, maxiter=200, complex_pairs=True, ops=NumpyOps(n)) for b in bs]
    sizes = []

    def block_matvec(vs):
        sizes.append(len(vs))
        return [A(v) for v in vs]

    block = gcrotmk_device_block(None, block_matvec, bs, n, rtol=1e-7, atol=1e-12, maxiter=200, complex_pairs=True,
                                 ops_factory=lambda: NumpyOps(n))
    for (x1, i1, s1), (xb, ib, sb) in zip(single, block):
        assert i1 == ib == 0 and s1 == sb
        np.testing.assert_array_equal(x1, xb)
    assert max(sizes) == len(bs) and min(sizes) >= 1 and sizes[0] == len(bs)       # blocks shrink as solves finish
    # a zero right-hand side and an immediately converged one drop out without a product
    res = gcrotmk_device_block(None, block_matvec, [np.zeros(n, dtype=complex), bs[0]], n, rtol=1e-7, atol=1e-12, maxiter=200,
                               complex_pairs=True, ops_factory=lambda: NumpyOps(n))
    assert res[0][1] == 0 and np.all(res[0][0] == 0) and res[1][1] == 0
    np.testing.assert_array_equal(res[1][0], single[0][0])

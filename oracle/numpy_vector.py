"""Oracle (test infrastructure): CPU restatement of the reference's ndarray backend.

``RefVector`` follows ``NumpyVector`` (numpyVector.py:23-238) method by method, quirks
included, so that the oracle loop in ``oracle.lanczos_ref`` behaves like the reference
on the same inputs:

* ctor writes solver defaults back into the caller's ``linearSystemArgs`` dict and all
  derived vectors share that dict (numpyVector.py:25-36, 58-64).
* ``vdot(conjugate=False)`` is the bilinear product (numpyVector.py:89-93).
* ``orthogonalize_against_set`` is ONE modified-Gram-Schmidt sweep with the bilinear
  product and the redundant division by q.q, returning ``None`` on lindep
  (numpyVector.py:121-145).
* ``solve`` builds ``sigma*x - H@x`` (or its negative) and calls SciPy ``minres`` /
  ``gcrotmk``; a non-zero ``info`` raises (numpyVector.py:147-178).  SciPy >= 1.14
  spells the tolerance ``rtol`` (the reference still passes ``tol=``): same value.
* Gram builders fill one triangle and mirror the conjugate (numpyVector.py:180-238).

Never imported by ``eigensolvers_amd``.
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse.linalg as spla
from scipy.sparse import csc_matrix

LINDEP_DEFAULT_VALUE = 1e-14      # abstractVector.py:12


class SolverNotConverged(UserWarning):
    """numpyVector.py:175-177 turns the warning into an exception; we raise directly."""


class RefVector:
    def __init__(self, array, options=None):
        self.array = array
        self.size = array.size
        self.shape = array.shape
        given = {} if options is None else options
        lsa = given.get("linearSystemArgs", dict())
        lsa.setdefault("linearSolver", "minres")
        lsa.setdefault("linearIter", 1000)
        lsa.setdefault("linear_tol", 1e-4)
        lsa.setdefault("linear_atol", 1e-4)
        self.options = {"linearSystemArgs": lsa}

    # -- properties (numpyVector.py:38-55)
    hasExactAddition = True
    maxD = 0

    @property
    def dtype(self):
        return self.array.dtype

    # -- out-of-place scaling (numpyVector.py:57-70)
    def __mul__(self, c):
        return RefVector(self.array * c, self.options)

    __rmul__ = __mul__

    def __truediv__(self, c):
        return RefVector(self.array / c, self.options)

    def __imul__(self, c):
        raise NotImplementedError

    def __itruediv__(self, c):
        raise NotImplementedError

    def __len__(self):
        return len(self.array)

    # -- norms / products (numpyVector.py:76-93)
    def norm(self):
        return sla.norm(self.array)            # scipy.linalg.norm (BLAS nrm2), as the reference

    def normalize(self):
        self.array /= sla.norm(self.array)
        return self

    def real(self):
        return RefVector(np.real(self.array), self.options)

    def conjugate(self):
        return RefVector(self.array.conj(), self.options)

    def vdot(self, other, conjugate=True):
        if conjugate:
            return np.vdot(self.array, other.array)
        return np.dot(self.array.ravel(), other.array.ravel())

    def copy(self):
        return RefVector(self.array.copy(), self.options)

    def applyOp(self, op):
        return RefVector(op @ self.array, self.options)      # numpyVector.py:98-100

    def compress(self):
        return self

    # -- static hooks ---------------------------------------------------------------
    @staticmethod
    def linearCombination(vectors, coeffs):                  # numpyVector.py:105-119
        assert len(vectors) == len(coeffs)
        acc = np.zeros(len(vectors[0]), dtype=vectors[0].dtype)
        for c, v in zip(coeffs, vectors):
            acc += c * v.array
        return RefVector(acc, vectors[0].options)

    @staticmethod
    def orthogonalize_against_set(x, qs, lindep=LINDEP_DEFAULT_VALUE):   # :121-145
        for q in qs:
            t1 = x.vdot(q, conjugate=False)
            t2 = q.vdot(q, conjugate=False)
            x = RefVector.linearCombination([x, q * (t1 / t2)], [1.0, -1.0])
        ip = x.vdot(x, conjugate=False)
        if ip > lindep:
            return x / np.sqrt(ip)
        return None

    @staticmethod
    def solve(H, b, sigma, x0=None, opType="her", reverseGF=False):       # :147-178
        n = H.shape[0]
        dtype = np.result_type(sigma, H.dtype, b.dtype)
        sgn = -1.0 if reverseGF else 1.0
        lin = spla.LinearOperator((n, n), dtype=dtype,
                                  matvec=lambda x: sgn * (sigma * x - H @ x))
        o = b.options["linearSystemArgs"]
        name = o["linearSolver"]
        if name == "gcrotmk":
            wk, info = spla.gcrotmk(lin, b.array, x0, rtol=o["linear_tol"],
                                    atol=o["linear_atol"], maxiter=o["linearIter"])
        elif name == "minres":
            wk, info = spla.minres(lin, b.array, x0, rtol=o["linear_tol"],
                                   maxiter=o["linearIter"])
        elif name == "pardiso":
            A1 = csc_matrix(sgn * (sigma * np.eye(n) - H))
            wk = spla.spsolve(A1, csc_matrix(np.reshape(b.array, (n, 1))))
            info = 0
        else:
            raise Exception("Got linear solver other than gcrotmk, minres and pardiso!")
        if info != 0:
            raise SolverNotConverged("Warning:: Iterative solver is not converged ")
        return RefVector(wk, b.options)

    @staticmethod
    def overlapMatrix(vectors):                              # numpyVector.py:192-203
        m = len(vectors)
        S = np.zeros((m, m), dtype=vectors[0].dtype)
        for i in range(m):
            for j in range(i, m):
                S[i, j] = vectors[i].vdot(vectors[j], True)
                S[j, i] = S[i, j].conj()
        return S

    @staticmethod
    def matrixRepresentation(op, vectors):                   # numpyVector.py:180-190
        m = len(vectors)
        M = np.zeros((m, m), dtype=vectors[0].dtype)
        for j in range(m):
            ket = vectors[j].applyOp(op)
            for i in range(j, m):
                M[i, j] = vectors[i].vdot(ket)
                M[j, i] = M[i, j].conj()
        return M

    @staticmethod
    def extendOverlapMatrix(vectors, S):                     # numpyVector.py:223-238
        m = len(vectors)
        row = np.empty((1, m), dtype=vectors[0].dtype)
        for i in range(m):
            row[0, i] = vectors[i].vdot(vectors[-1], True)
        S = np.append(S, row[:, :-1].conj(), axis=0)
        return np.append(S, row.T, axis=1)

    @staticmethod
    def extendMatrixRepresentation(op, vectors, M):          # numpyVector.py:205-221
        m = len(vectors)
        ket = vectors[-1].applyOp(op)
        row = np.empty((1, m), dtype=vectors[0].dtype)
        for i in range(m):
            row[0, i] = vectors[i].vdot(ket)
        M = np.append(M, row[:, :-1].conj(), axis=0)
        return np.append(M, row.T, axis=1)

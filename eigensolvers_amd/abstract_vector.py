"""The vector plugin surface the Lanczos loop is written against.

Same names, parameter names, defaults and return conventions as the reference's
``AbstractVector`` (abstractVector.py:15-169): three abstract properties, fourteen abstract
methods and eight static hooks whose default raises.  A backend is selected purely by
``type(v0[0])`` (inexact_Lanczos.py:284); the operator ``H`` is an opaque token that is only
handed back to the backend's own static hooks, so each backend defines what an operator is.

``tests/test_surface.py`` compares every signature below - and those of ``HipVector`` - with
``tests/golden/surface.json``, the ``inspect.signature`` dump of the reference's own classes
(``tests/golden/make_golden_r2.py``), so a drift from the reference's interface fails a test.
"""
from abc import ABC, abstractmethod

LINDEP_DEFAULT_VALUE = 1e-14          # abstractVector.py:12


class AbstractVector(ABC):
    """What the solvers may ask of a vector backend."""

    # ---- properties (abstractVector.py:17-37) ------------------------------------------
    @property
    @abstractmethod
    def hasExactAddition(self):
        """True when adding vectors is exact for this representation (c + c* == 2 Re c);
        tensor-network backends answer False because their sums are refitted."""

    @property
    @abstractmethod
    def dtype(self):
        """numpy dtype of the coefficients."""

    @property
    @abstractmethod
    def maxD(self):
        """Largest virtual bond dimension (tensor-network backends); 0 for dense storage."""

    # ---- arithmetic (abstractVector.py:39-61) -------------------------------------------
    @abstractmethod
    def __mul__(self, other):
        """Out-of-place scaling by a scalar: a new vector."""

    @abstractmethod
    def __rmul__(self, other):
        """scalar * vector: a new vector."""

    @abstractmethod
    def __truediv__(self, other):
        """Out-of-place division by a scalar: a new vector."""

    @abstractmethod
    def __imul__(self, other):
        """In-place scaling (the array backends raise NotImplementedError)."""

    @abstractmethod
    def __itruediv__(self, other):
        """In-place division (the array backends raise NotImplementedError)."""

    @abstractmethod
    def __len__(self):
        """Number of coefficients."""

    # ---- instance methods (abstractVector.py:63-97) --------------------------------------
    @abstractmethod
    def normalize(self):
        """Divide by the Euclidean norm in place; returns self."""

    @abstractmethod
    def norm(self):
        """Euclidean norm as a host float."""

    @abstractmethod
    def real(self):
        """Real part: a new vector."""

    @abstractmethod
    def conjugate(self):
        """Complex conjugate: a new vector."""

    @abstractmethod
    def vdot(self, other, conjugate=True):
        """<self|other> as a host scalar; self is conjugated unless conjugate=False."""

    @abstractmethod
    def copy(self):
        """Deep copy."""

    @abstractmethod
    def applyOp(self, other):
        """``other @ self`` as a new vector (``other`` is whatever the backend calls an operator)."""

    @abstractmethod
    def compress(self):
        """Compress the representation where that means something; may return self."""

    # ---- static hooks (abstractVector.py:99-169); the defaults raise ---------------------
    @staticmethod
    def linearCombination(other, coeff):
        """sum_n coeff[n] * other[n] as a new vector."""
        raise NotImplementedError("linearCombination")

    @staticmethod
    def orthogonalize(xs, lindep=LINDEP_DEFAULT_VALUE):
        """Orthonormalise a whole set of vectors."""
        raise NotImplementedError("orthogonalize")

    @staticmethod
    def orthogonalize_against_set(x, xs, lindep=LINDEP_DEFAULT_VALUE):
        """Orthonormalise x against the set xs; None when what is left of x has squared norm <= lindep."""
        raise NotImplementedError("orthogonalize_against_set")

    @staticmethod
    def solve(H, b, sigma, x0=None, opType="her", reverseGF=False):
        """x with (sigma*I - H) x = b, or (H - sigma*I) x = b when reverseGF; raises when the
        backend's iterative solver does not converge."""
        raise NotImplementedError("solve")

    @staticmethod
    def matrixRepresentation(operator, vectors):
        """Host m x m array <v_i| operator |v_j>."""
        raise NotImplementedError("matrixRepresentation")

    @staticmethod
    def overlapMatrix(vectors):
        """Host m x m array <v_i|v_j>."""
        raise NotImplementedError("overlapMatrix")

    @staticmethod
    def extendMatrixRepresentation(operator, vectors, opMat):
        """opMat grown by the row and column that belong to vectors[-1]."""
        raise NotImplementedError("extendMatrixRepresentation")

    @staticmethod
    def extendOverlapMatrix(vectors, overlap):
        """overlap grown by the row and column that belong to vectors[-1]."""
        raise NotImplementedError("extendOverlapMatrix")


PROPERTIES = ("hasExactAddition", "dtype", "maxD")
METHODS = ("__mul__", "__rmul__", "__truediv__", "__imul__", "__itruediv__", "__len__", "normalize", "norm",
           "real", "conjugate", "vdot", "copy", "applyOp", "compress")
STATIC_HOOKS = ("linearCombination", "orthogonalize", "orthogonalize_against_set", "solve", "matrixRepresentation",
                "overlapMatrix", "extendMatrixRepresentation", "extendOverlapMatrix")

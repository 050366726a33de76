"""The vector plugin surface the Lanczos loop is written against.

Same names, argument meaning and return conventions as the reference's
``AbstractVector`` (abstractVector.py:15-169): three properties, the arithmetic dunders,
eight instance methods and the eight static hooks.  A backend is selected purely by
``type(v0[0])`` (inexact_Lanczos.py:284); the operator ``H`` is an opaque token that is only
handed back to the backend's own static hooks, so each backend defines what an operator is.
"""
from abc import ABC, abstractmethod

LINDEP_DEFAULT_VALUE = 1e-14          # abstractVector.py:12


class AbstractVector(ABC):
    # ---- properties (abstractVector.py:17-37) ----
    @property
    @abstractmethod
    def hasExactAddition(self):
        """True when c + c* == 2 Re(c) holds exactly for this representation."""

    @property
    @abstractmethod
    def dtype(self):
        ...

    @property
    @abstractmethod
    def maxD(self) -> int:
        """Largest virtual bond dimension (tensor-network backends only; 0 otherwise)."""

    # ---- arithmetic (abstractVector.py:39-61) ----
    @abstractmethod
    def __mul__(self, other): ...

    @abstractmethod
    def __rmul__(self, other): ...

    @abstractmethod
    def __truediv__(self, other): ...

    @abstractmethod
    def __imul__(self, other): ...

    @abstractmethod
    def __itruediv__(self, other): ...

    @abstractmethod
    def __len__(self): ...

    # ---- instance methods (abstractVector.py:63-97) ----
    @abstractmethod
    def normalize(self):
        """Normalise in place and return self."""

    @abstractmethod
    def norm(self) -> float: ...

    @abstractmethod
    def real(self): ...

    @abstractmethod
    def conjugate(self): ...

    @abstractmethod
    def vdot(self, other, conjugate=True): ...

    @abstractmethod
    def copy(self): ...

    @abstractmethod
    def applyOp(self, other):
        """Return ``other @ self`` as a new vector."""

    @abstractmethod
    def compress(self):
        """Compress if compressible; may return self."""

    # ---- static hooks (abstractVector.py:99-169) ----
    @staticmethod
    def linearCombination(other, coeff):
        raise NotImplementedError

    @staticmethod
    def orthogonalize(xs, lindep=LINDEP_DEFAULT_VALUE):
        raise NotImplementedError

    @staticmethod
    def orthogonalize_against_set(x, xs, lindep=LINDEP_DEFAULT_VALUE):
        """Orthonormalise x against xs; ``None`` when x is linearly dependent on xs."""
        raise NotImplementedError

    @staticmethod
    def solve(H, b, sigma, x0=None, opType="her", reverseGF=False):
        """Solve (sigma*I - H) x = b   (reverseGF: (H - sigma*I) x = b)."""
        raise NotImplementedError

    @staticmethod
    def matrixRepresentation(operator, vectors):
        raise NotImplementedError

    @staticmethod
    def overlapMatrix(vectors):
        raise NotImplementedError

    @staticmethod
    def extendMatrixRepresentation(operator, vectors, opMat):
        raise NotImplementedError

    @staticmethod
    def extendOverlapMatrix(vectors, overlap):
        raise NotImplementedError

"""The vector plugin surface the Lanczos loop is written against.

Same names, argument meaning and return conventions as the reference's
``AbstractVector`` (abstractVector.py:15-169): three properties, the arithmetic dunders,
eight instance methods and the eight static hooks.  A backend is selected purely by
``type(v0[0])`` (inexact_Lanczos.py:284); the operator ``H`` is an opaque token that is only
handed back to the backend's own static hooks, so each backend defines what an operator is.

The interface is written down once as a table (name -> what the loop expects of it) and the
abstract base class is generated from it, so that the table is also what the tests and
``INTEGRATION.md`` enumerate.
"""
from abc import ABCMeta, abstractmethod

LINDEP_DEFAULT_VALUE = 1e-14          # abstractVector.py:12

# name -> contract.  Properties (abstractVector.py:17-37).
PROPERTIES = {
    "hasExactAddition": "True when c + c* == 2 Re(c) holds exactly for this representation",
    "dtype": "numpy dtype of the coefficients",
    "maxD": "largest virtual bond dimension (tensor-network backends only; 0 otherwise)",
}
# Operators and instance methods every backend must define (abstractVector.py:39-97).
METHODS = {
    "__mul__": "(other) out-of-place scaling, new vector",
    "__rmul__": "(other) scalar * vector, new vector",
    "__truediv__": "(other) out-of-place division by a scalar, new vector",
    "__imul__": "(other) in-place scaling (array backends raise NotImplementedError)",
    "__itruediv__": "(other) in-place division (array backends raise NotImplementedError)",
    "__len__": "() number of coefficients",
    "normalize": "() normalise in place and return self",
    "norm": "() Euclidean norm as a host float",
    "real": "() real part, new vector",
    "conjugate": "() complex conjugate, new vector",
    "vdot": "(other, conjugate=True) <self|other>, conjugating self unless told otherwise; host scalar",
    "copy": "() deep copy",
    "applyOp": "(other) ``other @ self`` as a new vector",
    "compress": "() compress if compressible; may return self",
}
# Static hooks; a backend that lacks one inherits a stub that raises (abstractVector.py:99-169).
STATIC_HOOKS = {
    "linearCombination": "(vectors, coeffs) sum_n coeffs[n] * vectors[n]",
    "orthogonalize": "(xs, lindep) orthonormalise a whole set",
    "orthogonalize_against_set": "(x, xs, lindep) orthonormalise x against xs; None when x is linearly dependent",
    "solve": "(H, b, sigma, x0=None, opType='her', reverseGF=False) solve (sigma*I - H) x = b, "
             "or (H - sigma*I) x = b with reverseGF",
    "matrixRepresentation": "(operator, vectors) <v_i| operator |v_j>, host m x m array",
    "overlapMatrix": "(vectors) <v_i|v_j>, host m x m array",
    "extendMatrixRepresentation": "(operator, vectors, opMat) append the row and column of vectors[-1]",
    "extendOverlapMatrix": "(vectors, overlap) append the row and column of vectors[-1]",
}


def _required(name, contract):
    def method(self, *args, **kwargs):
        raise NotImplementedError(name)
    method.__name__ = method.__qualname__ = name
    method.__doc__ = contract
    return abstractmethod(method)


def _stub(name, contract):
    def hook(*args, **kwargs):
        raise NotImplementedError(name)
    hook.__name__ = hook.__qualname__ = name
    hook.__doc__ = contract
    return staticmethod(hook)


def _build():
    body = {"__doc__": "Abstract vector backend; see PROPERTIES / METHODS / STATIC_HOOKS of this module.",
            "__module__": __name__}
    for name, contract in PROPERTIES.items():
        body[name] = property(_required(name, contract))
    for name, contract in METHODS.items():
        body[name] = _required(name, contract)
    for name, contract in STATIC_HOOKS.items():
        body[name] = _stub(name, contract)
    return ABCMeta("AbstractVector", (), body)


AbstractVector = _build()

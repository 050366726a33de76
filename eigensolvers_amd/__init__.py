"""eigensolvers_amd - MI355X-native backend for the inexact-Lanczos shift-and-invert path.

Public surface (mirrors the reference's module-level names for this path):

    from eigensolvers_amd import (AbstractVector, HipVector, HipCsrOperator, HipContext,
                                  inexactLanczosDiagonalization)

``HipVector`` plugs into ``inexactLanczosDiagonalization`` exactly where the reference's
``NumpyVector`` does; all arithmetic runs in hand-written gfx950 kernels behind the C ABI
of ``include/hipeig.h`` (``libhipeig.so``).  Importing this package does not touch the
GPU; creating a ``HipContext`` / ``HipVector`` does, and fails loudly without one.
"""
from .abstract_vector import AbstractVector, LINDEP_DEFAULT_VALUE
from .hip_vector import HipComplexVector, HipContext, HipCsrOperator, HipVector
from .checkpoint import latest_checkpoint, load_checkpoint, save_checkpoint
from .feast import feastDiagonalization
from .lanczos import inexactLanczosDiagonalization, KrylovSpace, true_residual_norms
from .subspace import (basisTransformation, find_nearest, get_pick_function_close_to_sigma,
                       get_pick_function_maxOvlp)

__all__ = ["AbstractVector", "LINDEP_DEFAULT_VALUE", "HipContext", "HipCsrOperator", "HipVector", "HipComplexVector",
           "feastDiagonalization", "latest_checkpoint", "load_checkpoint", "save_checkpoint",
           "inexactLanczosDiagonalization", "KrylovSpace", "true_residual_norms",
           "basisTransformation", "find_nearest", "get_pick_function_close_to_sigma",
           "get_pick_function_maxOvlp"]
__version__ = "0.1.0"

"""Host-side m x m subspace algebra of the Lanczos loop (m = Krylov dimension <= nBlock*L).

These stay on the host by design: the matrices are at most a few hundred square and the
work is LAPACK ``eigh``.  Each function states the reference routine whose result it
reproduces (util_funcs.py).
"""
import numpy as np
import scipy.linalg as sla

LINDEP_TOL = 1e-14


def loewdin_transform(S, tol=LINDEP_TOL):
    """S^(-1/2) restricted to the eigen-directions of S above ``tol``.

    util_funcs.py:233-247 (``lowdinOrtho``): returns ``(all_independent, X)`` with
    ``X = U[:, keep] * lam[keep]**-0.5``."""
    lam, U = sla.eigh(S)
    keep = lam > tol
    return bool(keep.all()), U[:, keep] * lam[keep] ** (-0.5)


def ritz_pairs(X, Hmat):
    """Eigen-decomposition of the projected operator in the orthonormalised basis
    (util_funcs.py:360-385, ``diagonalizeHamiltonian``): ``eigh(X^H Hmat X)``."""
    return sla.eigh(X.conj().T @ Hmat @ X)


def eigenvalue_change(ev, reference):
    """sum|reference - ev| / sum|ev| over the compared values (util_funcs.py:249-289)."""
    num = 0.0
    den = 0.0
    for a, b in zip(ev, reference):
        num += abs(b - a)
        den += abs(a)
    return num / den


def get_pick_function_close_to_sigma(sigma):
    """Order Ritz pairs by |theta - sigma| (util_funcs.py:330-344)."""
    def pick(transformMat, vectors, eigenvalues):
        return np.argsort(np.abs(eigenvalues - sigma))
    return pick


def get_pick_function_maxOvlp(reference_vector):
    """Order Ritz pairs by decreasing overlap with a reference vector (util_funcs.py:305-328)."""
    def pick(transformMat, vectors, eigenvalues):
        ov = np.array([v.vdot(reference_vector) for v in vectors], dtype=transformMat.dtype)
        return np.argsort(-np.abs(transformMat.conj().T @ ov))
    return pick


def find_nearest(array, value):
    """(index, element) of the entry closest to ``value`` (util_funcs.py:127-130)."""
    array = np.asarray(array)
    idx = int(np.abs(array - value).argmin())
    return idx, array[idx]


def basisTransformation(bases, coeffs):
    """New vectors Y C from the basis list (util_funcs.py:208-231).

    ``coeffs`` is an m-vector (one combination) or an m x k matrix (k combinations).  The
    reference's quirk for the trivial coefficient ``[1.0]`` - returning the basis list
    itself wrapped in a list - is preserved, since callers index ``[0]`` into the result."""
    cls = type(bases[0])
    coeffs = np.asarray(coeffs)
    if coeffs.ndim == 1:
        if len(coeffs) == 1 and coeffs[0] == 1.0:
            return [bases]
        return [cls.linearCombination(bases, coeffs)]
    block = getattr(cls, "linearCombinationBlock", None)
    if block is not None:
        return block(bases, coeffs)
    return [cls.linearCombination(bases, coeffs[:, j]) for j in range(coeffs.shape[1])]

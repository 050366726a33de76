"""Test-only row-partitioned ndarray backend over torch.distributed (gloo).

It mirrors what ``HipVector`` does with a communicator attached - every vector is the local
row slice, reductions all-reduce, operator applications all-gather the operand - but on
CPU, so that the N>1 behaviour of the host driver can be exercised without GPUs.  The
arithmetic comes from the oracle (``RefVector`` / ``minres_ref``); nothing here ships."""
import numpy as np
import torch
import torch.distributed as dist

from oracle.minres_ref import minres as minres_ref
from oracle.numpy_vector import RefVector


def _allsum(x):
    t = torch.tensor([float(x)], dtype=torch.float64)
    dist.all_reduce(t)
    return float(t[0])


def gdot(a, b):
    return _allsum(np.dot(a, b))


class SlabOperator:
    """Rows [begin,end) of a scipy CSR matrix; ``apply`` all-gathers the operand."""

    def __init__(self, H, ranges, rank):
        self.ranges = ranges
        b, e = ranges[rank]
        self.local = H[b:e]
        self.shape = H.shape
        self.dtype = H.dtype

    def apply(self, x_local):
        world = dist.get_world_size()
        stride = max(e - b for b, e in self.ranges)
        send = torch.zeros(stride, dtype=torch.float64)
        send[:len(x_local)] = torch.from_numpy(np.ascontiguousarray(x_local))
        parts = [torch.zeros(stride, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, send)
        full = np.concatenate([parts[r].numpy()[:e - b] for r, (b, e) in enumerate(self.ranges)])
        return self.local @ full


class DistRefVector(RefVector):
    def _wrap(self, arr):
        return DistRefVector(arr, self.options)

    def __mul__(self, c):
        return self._wrap(self.array * c)

    __rmul__ = __mul__

    def __truediv__(self, c):
        return self._wrap(self.array / c)

    def norm(self):
        return np.sqrt(gdot(self.array, self.array))

    def normalize(self):
        self.array /= self.norm()
        return self

    def vdot(self, other, conjugate=True):
        return gdot(self.array, other.array)

    def copy(self):
        return self._wrap(self.array.copy())

    def applyOp(self, op):
        return self._wrap(op.apply(self.array))

    @staticmethod
    def linearCombination(vectors, coeffs):
        r = RefVector.linearCombination(vectors, coeffs)
        return DistRefVector(r.array, vectors[0].options)

    @staticmethod
    def orthogonalize_against_set(x, qs, lindep=1e-14):
        for q in qs:
            x = DistRefVector.linearCombination([x, q * (x.vdot(q) / q.vdot(q))], [1.0, -1.0])
        ip = x.vdot(x)
        return x / np.sqrt(ip) if ip > lindep else None

    @staticmethod
    def solve(H, b, sigma, x0=None, opType="her", reverseGF=False):
        o = b.options["linearSystemArgs"]
        assert o["linearSolver"] == "minres"
        x, info, itn, istop = minres_ref(lambda v: sigma * v - H.apply(v), b.array, rtol=o["linear_tol"],
                                         maxiter=o["linearIter"], dot=gdot)
        if info != 0:
            raise UserWarning("Warning:: Iterative solver is not converged ")
        return DistRefVector(x, b.options)

    @staticmethod
    def overlapMatrix(vectors):
        m = len(vectors)
        S = np.zeros((m, m))
        for i in range(m):
            for j in range(i, m):
                S[i, j] = S[j, i] = vectors[i].vdot(vectors[j])
        return S

    @staticmethod
    def matrixRepresentation(op, vectors):
        m = len(vectors)
        M = np.zeros((m, m))
        for j in range(m):
            ket = vectors[j].applyOp(op)
            for i in range(j, m):
                M[i, j] = M[j, i] = vectors[i].vdot(ket)
        return M

    @staticmethod
    def extendOverlapMatrix(vectors, S):
        col = np.array([v.vdot(vectors[-1]) for v in vectors])
        S = np.append(S, col[None, :-1], axis=0)
        return np.append(S, col[:, None], axis=1)

    @staticmethod
    def extendMatrixRepresentation(op, vectors, M):
        ket = vectors[-1].applyOp(op)
        col = np.array([v.vdot(ket) for v in vectors])
        M = np.append(M, col[None, :-1], axis=0)
        return np.append(M, col[:, None], axis=1)

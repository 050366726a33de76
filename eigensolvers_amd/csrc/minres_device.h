// Scalar side of the device-resident MINRES (SciPy 1.15.3 scipy.sparse.linalg.minres, the routine
// the reference calls at numpyVector.py:163), shared by the single-vector driver (minres.hip) and
// the lock-step block driver (minres_block.hip) so that both evaluate the identical expressions.
#pragma once
#include <math.h>
#include "common.h"

#define MR_EPS 2.220446049250313e-16

struct MinresArgs {
  double sigma, sign, rtol;
  int maxiter;
  int nA, nC, nD;                 // number of values behind pA / pC / pD that add up to <v,y> / <y,y> / <x,x> (1 = the sum itself)
  const double* pA; const double* pC; const double* pD;
  int64_t sC;                     // stride between the nC values of pC: a row-partitioned run finds one share of <y,y> per rank in
                                  // the scalar slots that ride on the operand all-gather (GatherLayout::slot)
};

// Where a kernel that finishes its reduction in its last workgroup (common.h) leaves things: workgroup b stores its partial
// at part[b] (the host offsets `part` per sweep launch), the last of `tickets` workgroups adds base[0..count) and stores the
// total to tot (and to tot2 when non-null: the scalar slot of the operand exchange).
struct MinresRed {
  double* part;
  double* base;
  int count;
  unsigned tickets;
  unsigned* counter;
  double* tot;
  double* tot2;
};

__device__ __forceinline__ double sum_or_value(const double* p, int count, double* lds) {
  if (count == 1) return p[0];
  return block_sum_partials(p, count, lds);
}

// <y,y> of the previous KC: one value, or one share per rank (rank order, so every rank adds them alike)
__device__ __forceinline__ double minres_yy(const MinresArgs& a) {
  double bb = a.pC[0];
  for (int r = 1; r < a.nC; ++r) bb += a.pC[(int64_t)r * a.sC];
  return bb;
}

// Stopping tests of the iteration that has just completed (SciPy order).  S is a private copy.
__device__ __forceinline__ void minres_tests(MinresState& S, double xx, const MinresArgs& a) {
  if (S.itn == 0 || S.done) return;
  S.Anorm = sqrt(S.tnorm2);
  S.ynorm = sqrt(xx);
  const double epsx = S.Anorm * S.ynorm * MR_EPS;
  S.rnorm = S.phibar;
  S.test1 = (S.ynorm == 0.0 || S.Anorm == 0.0) ? INFINITY : S.rnorm / (S.Anorm * S.ynorm);
  S.test2 = (S.Anorm == 0.0) ? INFINITY : S.root / S.Anorm;
  S.Acond = S.gmax / S.gmin;
  int istop = S.pending_m1 ? -1 : 0;
  if (istop == 0) {
    const double t1 = 1.0 + S.test1, t2 = 1.0 + S.test2;
    if (t2 <= 1.0) istop = 2;
    if (t1 <= 1.0) istop = 1;
    if (S.itn >= a.maxiter) istop = 6;
    if (S.Acond >= 0.1 / MR_EPS) istop = 4;
    if (epsx >= S.beta1) istop = 3;
    if (S.test2 <= a.rtol) istop = 2;
    if (S.test1 <= a.rtol) istop = 1;
  }
  S.istop = istop;
  if (istop != 0) S.done = 1;
}


// Recurrences of one iteration once beta_{k+1}^2 = <y,y> is known (SciPy's order of evaluation).
__device__ __forceinline__ void minres_advance(MinresState& S, double bb) {
  S.oldb = S.beta;
  S.beta = sqrt(bb);
  S.tnorm2 += S.alfa * S.alfa + S.oldb * S.oldb + S.beta * S.beta;
  if (S.itn == 0 && S.beta / S.beta1 <= 10.0 * MR_EPS) S.pending_m1 = 1;
  S.oldeps = S.epsln;
  S.delta = S.cs * S.dbar + S.sn * S.alfa;
  S.gbar = S.sn * S.dbar - S.cs * S.alfa;
  S.epsln = S.sn * S.beta;
  S.dbar = -S.cs * S.beta;
  S.root = sqrt(S.gbar * S.gbar + S.dbar * S.dbar);
  S.gamma = fmax(sqrt(S.gbar * S.gbar + S.beta * S.beta), MR_EPS);
  S.cs = S.gbar / S.gamma;
  S.sn = S.beta / S.gamma;
  S.phi = S.cs * S.phibar;
  S.phibar = S.sn * S.phibar;
  S.denom = 1.0 / S.gamma;
  S.gmax = fmax(S.gmax, S.gamma);
  S.gmin = fmin(S.gmin, S.gamma);
  S.s = 1.0 / S.beta;
  S.itn += 1;
}

// Host-side initial record for a right-hand side of squared norm bb (> 0).
static inline void minres_init_state(MinresState* h, double bb) {
  memset(h, 0, sizeof(MinresState));
  h->beta1 = sqrt(bb);
  h->beta = h->beta1;
  h->phibar = h->beta1;
  h->cs = -1.0;
  h->gmin = 1.7976931348623157e308;
  h->s = 1.0 / h->beta;
}

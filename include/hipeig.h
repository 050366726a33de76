/*
 * hipeig.h - C ABI of libhipeig.so: the MI355X (gfx950) backend of the inexact-Lanczos
 * shift-and-invert hot path.
 *
 * This is the drop-in boundary.  The upstream reference has no FFI of its own: its
 * plugin surface is the Python ABC ``AbstractVector`` (abstractVector.py:15-169) and the
 * ndarray backend ``NumpyVector`` (numpyVector.py:23-238).  Every entry point below is
 * what a ``HipVector`` sitting beside ``NumpyVector`` binds through ``ctypes``; each one
 * cites the reference call site it replaces.  Signatures use plain pointers and sizes
 * only (no torch / numpy types).  All device pointers are raw ``double*`` obtained from
 * ``hipeig_vec_alloc``; a "vector" is the LOCAL row slice of the (possibly
 * row-partitioned) global vector, ``n`` always being the local length.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; ``hipeig_last_error()``
 *     returns a thread-local human-readable message for the last failure.
 *   - functions that return scalars to the host (dot, nrm2, gram, minres info ...) are
 *     synchronous on return; all others are asynchronous on the context's compute
 *     stream and ordered with respect to each other.
 *   - reductions use a fixed grid and a fixed summation tree: results are bitwise
 *     reproducible run to run for the same inputs, device and rank count.
 *   - with a communicator attached (hipeig_comm_init) every reduction is followed by an
 *     RCCL all-reduce and every operator application by an all-gather of x.
 */
#ifndef HIPEIG_H
#define HIPEIG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hipeig_ctx hipeig_ctx;   /* device, streams, workspaces, communicator      */
typedef struct hipeig_csr hipeig_csr;   /* device-resident sparse operator (local rows)    */

/* ---- context ---------------------------------------------------------------------- */
int hipeig_ctx_create(int device, hipeig_ctx** out);
int hipeig_ctx_destroy(hipeig_ctx* ctx);
int hipeig_ctx_sync(hipeig_ctx* ctx);                       /* wait for the compute stream */
const char* hipeig_last_error(void);
/* info[0]=CU count, [1]=wave size, [2]=total HBM bytes, [3]=free HBM bytes, [4]=L2 bytes  */
int hipeig_device_info(hipeig_ctx* ctx, int64_t info[8], char* name, int name_len);

/* ---- communicator (one process per GPU; RCCL over xGMI) --------------------------- */
/* The host side exchanges the 128-byte id out of band (eigensolvers_amd.distributed: a TCP exchange). */
int hipeig_comm_unique_id(void* id128);
/* path of the RCCL library in use (ROCm's /opt/rocm/lib/librccl.so unless HIPEIG_RCCL_LIB says otherwise) */
int hipeig_comm_library(char* path, int path_len);
int hipeig_comm_init(hipeig_ctx* ctx, int nranks, int rank, const void* id128);
int hipeig_comm_destroy(hipeig_ctx* ctx);
int hipeig_comm_info(hipeig_ctx* ctx, int* nranks, int* rank);
/* stats[0] = collectives (operand exchanges + all-reduces) issued by the most recent hipeig_minres / hipeig_minres_block call of
 * this rank: a row-partitioned MINRES iteration costs two - the exchange of the operand, which also carries every rank's
 * share of <y,y>, and ONE all-reduce carrying <v,y> and the previous iteration's <x,x>. */
int hipeig_comm_stats(hipeig_ctx* ctx, int64_t stats[4]);
/* Replica mode for work that is spread over ranks without partitioning the rows (FEAST: one contour
 * point per GPU, feast.py:186-201): partitioned = 0 keeps the communicator but switches the implicit
 * all-gather / all-reduce of every product / reduction off; the only exchange is then
 * hipeig_vec_allreduce (SUM of a whole vector over the ranks, in place), which replaces the serial
 * accumulation of the quadrature terms in updateQ (feast.py:105-121). */
int hipeig_comm_set_partitioned(hipeig_ctx* ctx, int partitioned);
int hipeig_vec_allreduce(hipeig_ctx* ctx, double* v, int64_t n);
/* Rehearsal backend: the same collectives between several contexts of ONE process (one host thread
 * per rank; host barrier + device-to-device copies, sums in rank order).  Lets the multi-rank path
 * be run on a single GPU, where RCCL refuses two ranks on one device.  Every rank's thread must
 * make the same sequence of calls; a rank waiting 120 s for its peers fails instead of hanging. */
/* ---- the operand exchange of a row-partitioned product (numpyVector.py:152: one H@x per MINRES iteration) ------------
 * Two backends behind the same calls: 0 = RCCL's ncclAllGather (default), 1 = direct peer writes - every rank stores its
 * slice into the gathered buffers of its peers over all xGMI links at once (csrc/comm_direct.hip).  The direct backend
 * needs the peers' buffers mapped: hipeig_direct_alloc returns this rank's 192-byte record (two hipIpc handles - gathered
 * buffers, arrival flags - and the PCI bus id of its device; capacity in doubles per buffer, at least the gathered length
 * of the largest operator; attach refuses when a peer's device is not visible or not addressable from here), the host
 * side all-gathers the records of all ranks in rank order (eigensolvers_amd.distributed) and hands them to
 * hipeig_direct_attach; hipeig_comm_set_gather_backend then switches (every rank at the same point, nothing in flight).
 * hipeig_comm_gather_info: info[0] backend, [1] attached, [2] capacity, [3] exchanges begun, [4] error word of the
 * bounded waits (0 = none), [5] HIPEIG_GATHER_CHUNKS override (0 = automatic).  A wait that gave up (HIPEIG_DIRECT_WAIT_S,
 * default 120 s: the skew between ranks it tolerates) makes the NEXT call that returns anything to the host - dot, nrm2,
 * the solvers, downloads, hipeig_ctx_sync - fail with rc 4; the error sticks to the context.                           */
int hipeig_comm_init_direct(hipeig_ctx* ctx, int nranks, int rank);   /* rank / size without RCCL: every exchange direct */
int hipeig_comm_set_allreduce_backend(hipeig_ctx* ctx, int backend);  /* small all-reduces: 0 RCCL, 1 the peers' mailboxes */
int hipeig_direct_alloc(hipeig_ctx* ctx, int64_t capacity_doubles, void* record192_out);
int hipeig_direct_attach(hipeig_ctx* ctx, const void* all_records /* nranks x 192 bytes */);
int hipeig_direct_release(hipeig_ctx* ctx);    /* frees the direct buffers again (after a collective fall-back to RCCL) */
int hipeig_comm_set_wait_limit(hipeig_ctx* ctx, double seconds);   /* limit of the bounded waits from now on; <= 0: default */
int hipeig_comm_set_gather_backend(hipeig_ctx* ctx, int backend);
int hipeig_comm_gather_info(hipeig_ctx* ctx, int64_t info[8]);
/* chunks of the operand exchange (1-4; 0 = automatic) for operators created from now on; the same on every rank */
int hipeig_comm_set_gather_chunks(hipeig_ctx* ctx, int nchunks);
/* measurement hooks: per-phase event times of the most recent partitioned product (ms; negative = phase absent):
 * out[0] operand exchange, [1] sweep of the rank's own column windows (under the exchange), [2] sweep of the other
 * windows incl. waits for later chunks, [3] whole product, [4] compute stream idle before the first chunk arrived;
 * and the average time of `reps` back-to-back all-reduces of `count` doubles.                                          */
int hipeig_phase_timing(hipeig_ctx* ctx, int on);
/* on = 0: products of this context place their own slice and SKIP the exchange (results meaningless) - times one rank's
 * sweeps alone on a shared GPU; not collective, switch back on before the next collective product                     */
int hipeig_comm_set_exchange(hipeig_ctx* ctx, int on);
int hipeig_phase_get(hipeig_ctx* ctx, double out[8]);
int hipeig_comm_bench_allreduce(hipeig_ctx* ctx, int count, int reps, double* ms_each);
int hipeig_loopback_group_create(int nranks, void** group_out);
int hipeig_loopback_group_destroy(void* group);
int hipeig_comm_init_loopback(hipeig_ctx* ctx, void* group, int rank);

/* ---- vectors: replaces the ndarray held by NumpyVector (numpyVector.py:25-28) ----- */
int hipeig_vec_alloc(hipeig_ctx* ctx, int64_t n, double** out);
int hipeig_vec_free(hipeig_ctx* ctx, double* v);
int hipeig_vec_upload(hipeig_ctx* ctx, double* dst, const double* host_src, int64_t n);
int hipeig_vec_download(hipeig_ctx* ctx, double* host_dst, const double* src, int64_t n);
int hipeig_vec_copy(hipeig_ctx* ctx, double* dst, const double* src, int64_t n);  /* copy(), :95 */
int hipeig_vec_fill(hipeig_ctx* ctx, double* v, int64_t n, double value);

/* ---- BLAS-1 class ----------------------------------------------------------------- */
/* vdot / np.dot, numpyVector.py:89-93 (real vectors: conjugated == bilinear)            */
int hipeig_dot(hipeig_ctx* ctx, int64_t n, const double* x, const double* y, double* out);
/* la.norm, numpyVector.py:80-81                                                          */
int hipeig_nrm2(hipeig_ctx* ctx, int64_t n, const double* x, double* out);
/* normalize(): x /= ||x||, numpyVector.py:76-78; returns the norm it divided by          */
int hipeig_normalize(hipeig_ctx* ctx, int64_t n, double* x, double* norm_out);
/* __mul__/__rmul__/__truediv__ (out of place), numpyVector.py:57-64: y = alpha*x         */
int hipeig_scale(hipeig_ctx* ctx, int64_t n, double alpha, const double* x, double* y);
/* y = x / alpha (a true division, as ndarray/alpha rounds)                               */
int hipeig_divide(hipeig_ctx* ctx, int64_t n, double alpha, const double* x, double* y);
/* y = a*x + b*y                                                                          */
int hipeig_axpby(hipeig_ctx* ctx, int64_t n, double a, const double* x, double b, double* y);
/* linearCombination, numpyVector.py:105-119: out = sum_j coeffs[j]*vecs[j]  (k >= 1)     */
int hipeig_lincomb(hipeig_ctx* ctx, int64_t n, int k, const double* coeffs,
                   const double* const* vecs, double* out);
/* basisTransformation, util_funcs.py:208-231: outs[c] = sum_j C[j*ldc + c]*vecs[j]       */
int hipeig_lincomb_block(hipeig_ctx* ctx, int64_t n, int m, int k, const double* C, int ldc,
                         const double* const* vecs, double* const* outs);

/* ---- tall-skinny products --------------------------------------------------------- */
/* out[j] = <Y_j, x>, j < m : one pass over x and the m basis columns.  Replaces the m
 * python-level vdots of extendOverlapMatrix / extendMatrixRepresentation
 * (numpyVector.py:205-238) and the projections of the Gram-Schmidt sweep (:132-139).     */
int hipeig_multi_dot(hipeig_ctx* ctx, int64_t n, int m, const double* const* Y,
                     const double* x, double* out);
/* x += sum_j c[j]*Y_j                                                                    */
int hipeig_multi_axpy(hipeig_ctx* ctx, int64_t n, int m, const double* const* Y,
                      const double* c, double* x);
/* out[i*mb + j] = <A_i, B_j> (row-major ma x mb). overlapMatrix (numpyVector.py:192-203)
 * with A == B; matrixRepresentation (:180-190) with B = H*A.                             */
int hipeig_gram(hipeig_ctx* ctx, int64_t n, int ma, const double* const* A, int mb,
                const double* const* B, double* out);
/* orthogonalize_against_set, numpyVector.py:121-145.  x is orthogonalised against the m
 * columns of Y in place and normalised by sqrt(x.x).  method 0 = the reference's single
 * modified-Gram-Schmidt sweep (division by q.q included), 1 = classical GS applied twice
 * (two batched passes, one reduction each).  *innerprod receives x.x BEFORE
 * normalisation; if it is <= lindep x is left un-normalised and *is_lindep = 1 (the
 * caller returns None, numpyVector.py:141-144).                                          */
int hipeig_orthonormalize(hipeig_ctx* ctx, int64_t n, int m, const double* const* Y,
                          double* x, double lindep, int method, double* innerprod,
                          int* is_lindep);

/* Sequential modified Gram-Schmidt projection, coefficients returned: for j = 0..m-1:
 * coeffs[j] = <V_j, w>; w -= coeffs[j]*V_j.  This is the Arnoldi orthogonalisation inside
 * scipy's gcrotmk/_fgmres, i.e. the inner loop of NumpyVector.solve with linearSolver="gcrotmk"
 * (numpyVector.py:161); no host round trip between columns.  The pair form treats (re, im)
 * buffer pairs as complex vectors with the conjugated product (complex contour solves of
 * feast.py:90); coeffs then holds m (re, im) pairs.                                       */
int hipeig_mgs_project(hipeig_ctx* ctx, int64_t n, int m, const double* const* V, double* w,
                       double* coeffs);
int hipeig_pair_mgs_project(hipeig_ctx* ctx, int64_t n, int m, const double* const* Vre,
                            const double* const* Vim, double* wre, double* wim, double* coeffs);
/* One whole Arnoldi step of scipy's _fgmres (the inner loop of gcrotmk, numpyVector.py:161) with a single
 * host round trip: out = [ ||w||^2 before, h_0..h_{m-1}, ||w||^2 after ]; w is orthogonalised against
 * V_0..V_{m-1} one column after the other and then scaled by 1/||w|| when that is finite.  The pair form
 * works on complex vectors held as (re, im) buffers: h_j = conj(V_j).w as (re, im), out has 2m + 2 doubles. */
int hipeig_arnoldi_step(hipeig_ctx* ctx, int64_t n, int m, const double* const* V, double* w, double* out);
int hipeig_pair_arnoldi_step(hipeig_ctx* ctx, int64_t n, int m, const double* const* Vre, const double* const* Vim,
                             double* wre, double* wim, double* out);

/* The same step with `cols_per_pass` columns per pass over w: 1 = the sequential sweep (scipy's order of rounding); 4 = a
 * pass applies four columns and forms the next four's products with the updated w together with their 4 x 4 Gram block,
 * from which the sequential coefficients follow exactly (h_k = <V_k,w> - sum_{l<k} h_l <V_k,V_l>): 2.5 instead of 4 vector
 * streams per column, a launch per four columns, coefficients equal to the sequential ones to rounding.            */
int hipeig_arnoldi_step_p(hipeig_ctx* ctx, int64_t n, int m, const double* const* V, double* w, double* out, int cols_per_pass);
int hipeig_pair_arnoldi_step_p(hipeig_ctx* ctx, int64_t n, int m, const double* const* Vre, const double* const* Vim,
                               double* wre, double* wim, double* out, int cols_per_pass);

/* Split form of the pair step for several independent steps in a row (the right-hand sides of a lock-step block solve):
 * begin enqueues the step and an asynchronous copy of its 2m + 2 scalars into pinned slot `slot` (0..15, 2m + 2 <= 126),
 * end waits for the stream and copies `count` of them out.  One GPU only.                                         */
int hipeig_pair_arnoldi_step_begin(hipeig_ctx* ctx, int64_t n, int m, const double* const* Vre, const double* const* Vim,
                                   double* wre, double* wim, int cols_per_pass, int slot);
/* `count` (<= 16) such steps in ONE launch, a workgroup per step, for lengths at which a step is a single workgroup
 * (n <= 8192): step i has m[i] columns Vre[64 i + j], Vim[64 i + j], works on (wre[i], wim[i]) and reports into slot i
 * (collect with hipeig_arnoldi_step_end).  Returns 5, having done nothing, for longer vectors (take the step-by-step form). */
int hipeig_pair_arnoldi_step_batch_begin(hipeig_ctx* ctx, int64_t n, int count, const int* m, const double* const* Vre,
                                         const double* const* Vim, double* const* wre, double* const* wim);
int hipeig_arnoldi_step_end(hipeig_ctx* ctx, int slot, int count, double* out);

/* ---- sparse operator: replaces the scipy.sparse / ndarray H handed to the loop ----- */
/* Host CSR -> device.  rowptr has nrows+1 entries (local rows), col holds GLOBAL column
 * indices in [0, ncols).  Rows may be empty or unsorted; duplicates are summed by the
 * product.  row_offset = global index of local row 0 (row partition), used for the fused
 * shift term sigma*x[row].                                                               */
int hipeig_csr_create(hipeig_ctx* ctx, int64_t nrows, int64_t ncols, int64_t row_offset,
                      const int64_t* rowptr, const int32_t* col, const double* val,
                      hipeig_csr** out);
/* Seeded synthetic "gapped random-sparse Hermitian" operator generated on the device
 * (rows [row_begin,row_end) of the N x N matrix); bit-identical to
 * eigensolvers_amd.generators.gapped_csr_host.  params: see generators.py.               */
int hipeig_csr_generate(hipeig_ctx* ctx, int64_t N, int64_t row_begin, int64_t row_end,
                        int K, uint64_t seed, double eps, uint32_t keep_thresh24,
                        const double* targets, int ntargets, hipeig_csr** out);
int hipeig_csr_destroy(hipeig_ctx* ctx, hipeig_csr* A);
/* info[0]=nrows [1]=ncols [2]=nnz [3]=row_offset [4]=kernel variant of the last launch
 * (1 CSR-vector, 2 CSR-stream, 3/4 column-window blocked with wave / workgroup units)
 * [5]=device bytes [6]=row blocks [7]=kernel launches (sweeps) per product of that variant */
int hipeig_csr_info(hipeig_csr* A, int64_t info[8]);
/* layout constants of the blocked copy the last product ran on: out[0] variant, [1] rows per row block, [2] window bits,
 * [3] row blocks, [4] windows, [5] column splits, [6] workgroups per sweep launch, [7] threads per workgroup, [8] batch
 * unroll, [9] chunks of the operand exchange, [10] rows per (rank, chunk) of the gathered layout (0: not partitioned),
 * [11] log2 of the columns per bin + 100 when bins are aligned to 64-element instruction groups                  */
int hipeig_csr_layout_info(hipeig_csr* A, int64_t out[12]);
/* copy the device CSR (local rows) back to the host; pass NULL to skip an array          */
int hipeig_csr_download(hipeig_ctx* ctx, hipeig_csr* A, int64_t* rowptr, int32_t* col,
                        double* val);
/* force a kernel variant (0 = automatic choice, 1..4 as above, 5 = variant 4 with fixed-point (int64) LDS
 * accumulators: bitwise reproducible whatever the order in which waves reach a row, same speed; absolute error per
 * row ~ nnz_row * 2^-61 * max_i sum_j|a_ij| * max|x|) */
int hipeig_csr_set_variant(hipeig_csr* A, int variant);
/* options["reduction"] = "deterministic" of the vectors: on != 0 restricts the AUTOMATIC choice
 * (variant 0) to bitwise reproducible kernels - CSR-stream where it would be picked anyway, variant 5 in place of 4,
 * the row-owner kernel for block products.  A pinned variant is not touched.                                      */
int hipeig_csr_set_reproducible(hipeig_csr* A, int on);
/* variant 5's error-bound ingredients: out[0] = max_i sum_j |a_ij| over the local rows, out[1] = max |x| of the
 * operand of the most recent variant-5 product; absolute error per row <= (nnz_row/2 + 1) * 2^-60 * out[0] * out[1] */
int hipeig_csr_fixed_info(hipeig_ctx* ctx, hipeig_csr* A, double out[2]);

/* applyOp, numpyVector.py:98-100: y = H x.  x,y are local slices.                        */
int hipeig_spmv(hipeig_ctx* ctx, hipeig_csr* A, const double* x, double* y);
/* the LinearOperator lambda of solve(), numpyVector.py:152/154:
 * y = sign*(sigma*x - H x), sign = +1 (Green's function) or -1 (reverseGF)               */
int hipeig_spmv_shift(hipeig_ctx* ctx, hipeig_csr* A, double sigma, double sign,
                      const double* x, double* y);
/* The same lambda for a COMPLEX shift and operand (feast.py:83-90 -> numpyVector.py:152-161 with sigma = z_k on the
 * contour; H real): y = sign*(z*x - H x) with x = xr + i xi, z = zr + i zi, y = yr + i yi, the halves in separate
 * buffers.  sign = 0: the plain product y = H x (applyOp on a complex vector, numpyVector.py:98-100).  On a large
 * unpartitioned operator both halves share ONE sweep of the (index, value) stream; otherwise two sweeps.
 * hipeig_csr_pair_info: out[0] = 1 when the most recent call took the one-sweep form, out[1] = its launches.       */
int hipeig_spmv_shift_pair(hipeig_ctx* ctx, hipeig_csr* A, double zr, double zi, double sign,
                           const double* xr, const double* xi, double* yr, double* yi);
int hipeig_csr_pair_info(hipeig_csr* A, int64_t out[2]);

/* The k applications of matrixRepresentation (numpyVector.py:184-185) as ONE block product:
 * Y[j] = H X[j], j < k.  Internally the operands are interleaved so that each non-zero costs
 * one index fetch and one contiguous gather for all k (tall-skinny SpMM).                   */
int hipeig_spmm(hipeig_ctx* ctx, hipeig_csr* A, int k, const double* const* X, double* const* Y);
/* The complex matvec of the contour solves for SEVERAL right-hand sides (feast.py:198-200: the m0 solves of a contour point
 * share operator and shift): Y_p = sign*(z*X_p - H X_p), p < npairs, the complex operands as (re, im) buffers.  Four operands
 * share one pass over the operator (an 8-wide block product whose epilogue applies the shift).                        */
int hipeig_spmm_shift_pairs(hipeig_ctx* ctx, hipeig_csr* A, int npairs, double zr, double zi, double sign,
                            const double* const* Xre, const double* const* Xim, double* const* Yre, double* const* Yim);
/* Kernel choice of the block product / block solve: 0 = automatic, 1 = row-owner CSR (a wavefront per row,
 * 64-byte gathers from wherever the operand block lives), 2 = column-window blocked (operand windows
 * L2-resident, 8 accumulators per row in LDS).  info[0] = variant of the last block launch, [1] = row
 * blocks, [2] = column windows, [3] = rows per row block of the window-blocked layout.              */
int hipeig_csr_set_block_variant(hipeig_csr* A, int variant);
int hipeig_csr_block_info(hipeig_csr* A, int64_t info[4]);

/* ---- inner linear solve: NumpyVector.solve with linearSolver="minres" (:147-178) ---- */
/* Solves sign*(sigma*I - H) x = b from a zero initial guess with the Paige-Saunders
 * MINRES recurrences in the evaluation order of scipy.sparse.linalg.minres (the call at
 * numpyVector.py:163), all vectors and recurrence scalars device-resident.
 * out_stats: [0]=iterations [1]=istop (SciPy's code) [2]=rnorm estimate [3]=Anorm
 *            [4]=ynorm [5]=test1 [6]=test2 [7]=Acond.
 * *info follows SciPy: maxiter if the iteration limit was hit (istop==6) else 0.         */
int hipeig_minres(hipeig_ctx* ctx, hipeig_csr* A, double sigma, double sign, const double* b,
                  double* x, double rtol, int maxiter, int* info, double out_stats[8]);
/* The same with SciPy's initial guess (numpyVector.py:163 hands x0 on to scipy.sparse.linalg.minres): r1 = b - A x0,
 * the iterate starts at x0.  x0 == NULL is hipeig_minres.  The reference's solvers never pass one
 * (inexact_Lanczos.py:99, feast.py:90-97); the parameter is part of NumpyVector.solve's signature.            */
int hipeig_minres_x0(hipeig_ctx* ctx, hipeig_csr* A, double sigma, double sign, const double* b,
                     const double* x0, double* x, double rtol, int maxiter, int* info,
                     double out_stats[8]);

/* linearSolver = "pardiso" (numpyVector.py:166-170: spsolve of sigma*I - H, kept by the reference "only for comparing
 * with fortran" - the 4 x 4 known-answer system of unittests/test_feast_fortran.py): x = (sign*(z I - H))^-1 b by
 * Gaussian elimination with partial pivoting in the LDS of one workgroup, n <= 96, z = zr + i zi.  b_im and x_im may be
 * NULL for a real system (zi == 0).  *singular != 0: a zero pivot column, x is not written.                    */
int hipeig_dense_solve_small(hipeig_ctx* ctx, hipeig_csr* A, double zr, double zi, double sign,
                             const double* b_re, const double* b_im, double* x_re, double* x_im,
                             int* singular);

/* The nBlock solves of one block-Lanczos iteration (inexact_Lanczos.py:319-320: one NumpyVector.solve
 * per block vector, same operator, same shift) advanced in lock step: k <= 8 right-hand sides, ONE block
 * product per iteration.  Column j runs exactly the recurrences and stopping tests of hipeig_minres on
 * b[j]; a column that has stopped is masked, so x[j] is the iterate SciPy would return for it.
 * info[j] as in hipeig_minres; out_stats (may be NULL) receives k records of 8 doubles.             */
int hipeig_minres_block(hipeig_ctx* ctx, hipeig_csr* A, double sigma, double sign, int k,
                        const double* const* b, double* const* x, double rtol, int maxiter,
                        int* info, double* out_stats);

/* ---- timing on the library's compute stream (HIP events) --------------------------- */
int hipeig_timer_start(hipeig_ctx* ctx);
int hipeig_timer_stop(hipeig_ctx* ctx, float* elapsed_ms);   /* synchronous */

#ifdef __cplusplus
}
#endif
#endif /* HIPEIG_H */

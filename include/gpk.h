/* gpk.h -- C ABI of libgpk.so: the MI355X (gfx950) implementation of the Gram-assembly + Gauss-Newton hot path of
 * yifanc96/NonLinPDEs-GPsolver.  The reference has no FFI of its own (it is pure Python on JAX); the boundary this
 * library replaces is the set of array-level operations its src/ modules hand to XLA.  Each entry point names the
 * reference call site(s) it stands in for (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - return int: 0 ok; >0 LAPACK-style info (1-based index of the first non-positive pivot); <0 = -(hipError_t),
 *     text via gpk_last_error().  -9001 = invalid argument, -9002 = library built without a usable device.
 *   - every matrix/vector pointer is a DEVICE pointer owned by the caller unless the name says `host`; the library
 *     never frees or keeps them after the call.  Matrices are row-major double with a leading dimension in ELEMENTS;
 *     symmetric outputs are stored full.  Vector loads are 16-byte wide when pointer and leading dimension allow
 *     (even ld, 16-byte aligned base); any alignment is accepted.
 *   - calls are asynchronous on the handle's stream; functions that return host scalars (info, loss, ratios)
 *     synchronise that stream.  One host thread per handle.  The library has NO process-wide mutable state (round 4): all
 *     state, including the development / tuning switches (gpk_debug.h: gpk_tune(handle, key, value), never called by the
 *     product path), lives in the handle, so handles of one process are independent of each other.
 *   - there is NO CPU fallback: without a gfx950 device gpk_create fails.
 */
#ifndef GPK_H
#define GPK_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpk_ctx* gpk_handle;

/* Gram layouts = the `eqn` strings of src/Gram_matrice.py:41,58,100,137 (Darcy yields two matrices: _U and _A). */
enum { GPK_LAYOUT_ELLIPTIC = 0, GPK_LAYOUT_BURGERS = 1, GPK_LAYOUT_EIKONAL = 2, GPK_LAYOUT_DARCY_U = 2, GPK_LAYOUT_DARCY_A = 3 };
/* kernel classes of src/kernels.py:8,91 */
enum { GPK_KERNEL_GAUSSIAN = 0, GPK_KERNEL_ANISOTROPIC = 1 };
/* nugget_type of src/PDEs.py:56,250,391 / src/InverseProblems.py:66 */
enum { GPK_NUGGET_NONE = 0, GPK_NUGGET_IDENTITY = 1, GPK_NUGGET_ADAPTIVE = 2 };
/* Gauss-Newton systems (measurement vector F(z) and Jacobian A(z) of each equation class) */
enum { GPK_GN_ELLIPTIC = 0, GPK_GN_BURGERS = 1, GPK_GN_EIKONAL = 2, GPK_GN_DARCY = 3, GPK_GN_ELLIPTIC_RELAXED = 4 };

/* ---- context, memory, stream ------------------------------------------------------------------------------ */
int gpk_create(int device, gpk_handle* out);
int gpk_destroy(gpk_handle h);
const char* gpk_last_error(gpk_handle h);
const char* gpk_version(void);
int gpk_set_stream(gpk_handle h, void* hip_stream);          /* NULL = the handle's own stream */
int gpk_synchronize(gpk_handle h);
int gpk_device_info(gpk_handle h, char* name, int name_len, int* compute_units, size_t* hbm_bytes, int* clock_khz);
int gpk_malloc(gpk_handle h, size_t bytes, void** dptr);
int gpk_free(gpk_handle h, void* dptr);
int gpk_memset(gpk_handle h, void* dptr, int value, size_t bytes);
int gpk_memcpy_h2d(gpk_handle h, void* dst, const void* host_src, size_t bytes);
int gpk_memcpy_d2h(gpk_handle h, void* host_dst, const void* src, size_t bytes);
int gpk_memcpy_d2d(gpk_handle h, void* dst, const void* src, size_t bytes);
int gpk_memcpy2d_h2d(gpk_handle h, void* dst, size_t dpitch, const void* host_src, size_t spitch, size_t width, size_t height);
int gpk_memcpy2d_d2h(gpk_handle h, void* host_dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height);
int gpk_memcpy2d_d2d(gpk_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height);

/* HIP-event stopwatch on the handle's stream (used by bench.py for per-kernel durations). */
int gpk_timer_start(gpk_handle h);
int gpk_timer_stop(gpk_handle h, double* host_ms);            /* synchronises */

/* Per-phase HIP-event timing inside gpk_gn_step, recorded on the handle's stream (bench.py's roofline leg):
 * host_ms4 = accumulated milliseconds of {TRSM phase, the SYRK launch, POTRF of H, TRSV + update} over *host_count steps. */
int gpk_prof_enable(gpk_handle h, int on);                    /* also resets the accumulators */
int gpk_prof_read(gpk_handle h, double* host_ms4, int* host_count);
/* gpk_gn_step forms Hb = S^T S and factors it in one pipelined phase (product by 512-column blocks on a GEMM stream, panel
 * chains on a second stream with a disjoint CU mask): host_ms4[1] is then the wall time of that whole phase and host_ms4[2]
 * is 0.  *host_pipelined = 1 if the last step ran pipelined; *host_syrk_launch_ms = accumulated duration of the SYRK launches
 * themselves (HIP events on the stream they ran on); *host_chain_cus = CUs of the chain partition. */
int gpk_prof_read_pipeline(gpk_handle h, int* host_pipelined, double* host_syrk_launch_ms, int* host_chain_cus);
/* Flops EXECUTED by the matrix-product launches issued while the per-phase timing was on, accumulated per phase from the K ranges
 * the launch logic gives its tiles (structural zeros of A(z), triangular operands and skipped upper tiles left out):
 * host_flops4 = {solve S = L^{-1}[A | F], updates inside the factorisation of Hb, the product Hb = S^T S, unused};
 * host_launches4 (may be NULL) = number of launches behind each figure.  The panel / substitution kernels are not counted. */
int gpk_prof_read_flops(gpk_handle h, double* host_flops4, long* host_launches4);
/* Duration of the Gram evaluator launch of the LAST gpk_assemble call issued while the per-phase timing was on (HIP events on the
 * handle's stream around that launch alone; the point packing kernel and the host-side set-up stay outside).  Synchronises. */
int gpk_prof_read_assembly(gpk_handle h, double* host_ms);

/* ---- Gram assembly: replaces Gram_matrix_assembly (src/Gram_matrice.py:11-187) plus the nugget of
 *      *.Gram_matrix (src/PDEs.py:56-73,250-269,391-409; src/InverseProblems.py:66-99) in one fused pass.
 *      kparams: Gaussian {sigma, unused}; anisotropic {sigma_t, sigma_x}.  Xd (Nd,2), Xb (Nb,2) row-major.
 *      Theta: N x N, N = 2Nd+Nb (ELLIPTIC), 4Nd+Nb (BURGERS/EIKONAL/DARCY_U), 3Nd (DARCY_A).
 *      host_ratios[3]: adaptive trace ratios (unused entries 0). */
int gpk_assemble(gpk_handle h, int layout, int kernel, const double* host_kparams,
                 const double* Xd, int Nd, const double* Xb, int Nb,
                 double nugget, int nugget_type, double* Theta, int ld, double* host_ratios);
/* construct_Theta_test (src/Gram_matrice.py:190-289): out is Nt x N row-major. */
int gpk_assemble_test(gpk_handle h, int layout, int kernel, const double* host_kparams,
                      const double* Xt, int Nt, const double* Xd, int Nd, const double* Xb, int Nb,
                      double* out, int ld);
/* Theta_test @ coeff without materialising Theta_test: the matmul of *.extend_sol
 * (src/PDEs.py:208,350,505; src/InverseProblems.py:193,196).  coeff (N,), out (Nt,). */
int gpk_extend(gpk_handle h, int layout, int kernel, const double* host_kparams,
               const double* Xt, int Nt, const double* Xd, int Nd, const double* Xb, int Nb,
               const double* coeff, double* out);
/* solver_GP.collocation_pts_err / get_test_error (src/solver.py:169-178, 185-194): err_all[i] = |truth[i] - approx[i]| (device, may be
 * NULL), *host_max = max_i err_all[i], *host_l2 = sqrt(sum_i err_all[i]^2 / n) -- the reference's "L2 error".  All inputs on the
 * device (the extension already is); one pass, fixed summation order.  Synchronises. */
int gpk_error_metrics(gpk_handle h, int n, const double* truth, const double* approx, double* err_all,
                      double* host_max, double* host_l2);

/* ---- dense fp64 linear algebra on the MFMA units --------------------------------------------------------- */
/* jnp.linalg.cholesky (src/PDEs.py:77,273,413; src/InverseProblems.py:102-103): lower factor in place (strict
 * upper triangle left untouched -- gpk_tril zeroes it).  Non-positive pivot: *host_info = its 1-based index,
 * NaNs propagate like the reference's JAX path; return value 0 unless a HIP error occurred. */
int gpk_potrf(gpk_handle h, double* A, int n, int lda, int* host_info);
int gpk_tril(gpk_handle h, double* A, int n, int lda);
int gpk_symmetrize_lower(gpk_handle h, double* A, int n, int lda);   /* copy lower triangle into the upper */
/* jnp.linalg.solve(self.L, .) with the triangular factor (src/PDEs.py:86,97,143,161,288,306,429,450;
 * src/InverseProblems.py:118-119,145-146): B <- L^{-1} B (trans=0) or L^{-T} B (trans=1); B is n x nrhs. */
int gpk_trsm(gpk_handle h, int trans, const double* L, int n, int ldl, double* B, int nrhs, int ldb);
/* Tall block column A (nrows x ncols, nrows >= ncols <= 512 recommended): the top ncols x ncols block is replaced by its
 * Cholesky factor L (lower) and the rows below by A[ncols:, :] L^{-T} -- one panel step of the blocked factorisation
 * (jnp.linalg.cholesky, src/PDEs.py:77), the owner's share of a step of the panel-sharded multi-GPU factorisation.
 * host_info as gpk_potrf (may be NULL: no host synchronisation then). */
int gpk_potrf_panel(gpk_handle h, double* A, int nrows, int ncols, int lda, int* host_info);
/* The same panel step for schedules that must not touch the host per panel (gpk/sharded.py, look-ahead): the pivot status
 * accumulates in the handle's device-side info word (first failure wins; pivot indices offset by pivot_base = the panel's first
 * column in the whole matrix).  gpk_info_reset before the first panel, ONE gpk_info_read (synchronises) after the last. */
int gpk_potrf_panel_at(gpk_handle h, double* A, int nrows, int ncols, int lda, int pivot_base);
int gpk_info_reset(gpk_handle h);
int gpk_info_read(gpk_handle h, int* host_info);
/* X <- X L^{-T} (X is m x n): the panel solve of the blocked Cholesky; exported for the multi-GPU panel-sharded
 * factorisation, whose per-panel schedule lives in the host layer (gpk/sharded.py). */
int gpk_trsm_right_lt(gpk_handle h, const double* L, int n, int ldl, double* X, int m, int ldx);
/* B <- L^{-T} L^{-1} B (src/PDEs.py:205,347,502; src/InverseProblems.py:190,195) */
int gpk_potrs(gpk_handle h, const double* L, int n, int ldl, double* B, int nrhs, int ldb);
/* C <- alpha*op(A)*op(B) + beta*C; ta/tb = 1 means the operand is stored transposed (op(A) is m x k). */
int gpk_gemm(gpk_handle h, int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda,
             const double* B, int ldb, double beta, double* C, int ldc);
/* Leading-zero variants used by the column-sharded Gauss-Newton step (gpk/sharded.py) on the [A(z) | F] block built by
 * gpk_gn_build_rev: column c < lead of the right-hand side / of operand B is zero above row lead-1-c, columns >= lead are
 * dense.  Same results as gpk_trsm / gpk_gemm(tb = 0); the zeros are skipped (contiguous active column ranges in the
 * solve, late K start per tile in the product).  lead <= 0 degenerates to the dense routines. */
int gpk_trsm_lz(gpk_handle h, const double* L, int n, int ldl, double* B, int nrhs, int ldb, int lead);
int gpk_gemm_lz(gpk_handle h, int ta, int m, int n, int k, double alpha, const double* A, int lda,
                const double* B, int ldb, double beta, double* C, int ldc, int lead);
/* Companion of a Cholesky factor that is reused for many multi-right-hand-side forward solves (the factor of Theta in
 * GN_method: jnp.linalg.solve(self.L, .) inside every Hessian_GN, src/PDEs.py:97,306,450; src/InverseProblems.py:145-146):
 * Dinv (n x block doubles, leading dimension block; block = 256, 512, 1024 or 2048) receives the inverses of the block x block
 * diagonal blocks of L, block k in rows [k block, k block + n_k), computed by substitution.  gpk_trsm_dinv then solves
 * L X = B with GEMMs only: X (n x nrhs, ld ldx, must not alias B) receives L^{-1} B, B is overwritten with intermediate
 * values.  lead > 0: column c < lead of B is zero above row lead-1-c (as gpk_trsm_lz); the zero part of X left of that
 * boundary is NOT written, so X must be zero there on entry (zero X once; solves of the same shape keep it valid).
 * Accuracy: DESIGN.md section 4 (Gauss-Newton iterates within 3e-13 of the substitution path at nugget 1e-13, config 2). */
int gpk_trtri_diag(gpk_handle h, const double* L, int n, int ldl, double* Dinv, int block);
int gpk_trsm_dinv(gpk_handle h, const double* L, const double* Dinv, int block, int n, int ldl, double* B, int nrhs, int ldb,
                  double* X, int ldx, int lead);
/* C <- alpha*A^T A + beta*C, A is k x n row-major; lower triangle only unless full != 0
 * (the 2*ss^T ss of src/PDEs.py:307 and of every autodiff Hessian_GN). */
int gpk_syrk(gpk_handle h, int n, int k, double alpha, const double* A, int lda, double beta, double* C, int ldc, int full);

/* ---- Gauss-Newton ------------------------------------------------------------------------------------------ */
typedef struct {
    int system;              /* GPK_GN_* */
    int Nd, Nb, Ndata;
    double p0, p1;           /* ELLIPTIC(_RELAXED): alpha, m   BURGERS: alpha, nu   EIKONAL: eps, -   DARCY: noise_level, - */
    double pen_lambda;       /* ELLIPTIC_RELAXED only */
    const double* rhs_f;     /* (Nd,)  */
    const double* bdy_g;     /* (Nb,)  */
    const double* data_u;    /* (Ndata,) DARCY only */
    const double* L;  int ldl;     /* factor of Theta (DARCY: L_u) */
    const double* L2; int ldl2;    /* DARCY: L_a */
    const double* Dinv;            /* optional (may be NULL): gpk_trtri_diag(L)  -- the S solve then runs as GEMMs only */
    const double* Dinv2;           /* optional, DARCY: gpk_trtri_diag(L2) */
    int dinv_block;                /* block size Dinv / Dinv2 were built with (0 = 256) */
    /* optional (may be NULL) -- see gpk_gn_structured_prepare (v0: GPK_GN_ELLIPTIC only) */
    const double* W1; const double* W2; const double* v0; int ldw;
    /* optional (may be NULL), needs W1/W2/v0 -- see gpk_gn_gram_prepare */
    const double* G; int ldg; const double* pvec;
    /* optional (may be NULL), GPK_GN_DARCY only -- see gpk_gn_darcy_prepare */
    const double* Wa; int ldwa; const double* Ha; int ldha;
} gpk_gn_problem;

/* sizes: nz unknowns, rows of the stacked S = [L^{-1}A | L^{-1}F] buffer */
int gpk_gn_dims(const gpk_gn_problem* host_prob, int* nz, int* s_rows);
/* Workspace query (pure host function, no device): what a caller has to allocate for gpk_gn_step / gpk_gn_hessian_grad and what the
 * handle will reserve by itself.  lds: the leading dimension the caller intends to use for S and Hb (0 = the smallest admissible one,
 * nz + 1 rounded up to 16 doubles; returned in *host_lds).  *S_bytes = s_rows * lds * 8, *Hb_bytes = (nz + 1) * lds * 8, *delta_bytes =
 * nz * 8, *handle_bytes = the out-of-place solve buffer the handle grows on first use when the inverted diagonal blocks of EVERY factor are
 * supplied (s_rows * lds * 8 -- whatever dinv_block says, 0 meaning 256 -- else 0; + s_rows * 8 for the exact in-step loss).  An UPPER
 * BOUND on the large buffers (the loss vector is counted whether or not gpk_tune key 52 is on); on top of it the handle keeps two sets
 * of hand-off records of its single-vector solves (2 x 4 MB, fixed) once the in-step loss runs on its side stream.  Any output pointer may be NULL. */
int gpk_gn_worksize(const gpk_gn_problem* host_prob, int lds, int* host_lds, size_t* S_bytes, size_t* Hb_bytes, size_t* delta_bytes,
                    size_t* handle_bytes);
/* (With host_prob->Dinv set, S is scratch and the solved block lives in the handle's workspace.) */
/* One Gauss-Newton step = Hessian_GN + grad_loss + linear solve + update (src/PDEs.py:117-119,322-325,472-475;
 * src/InverseProblems.py:164-166) and the loss of the INPUT iterate (src/PDEs.py:82-87 ...):
 *   S  = [L^{-1}A(z) | L^{-1}F(z)]                (s_rows x (nz+1), ld lds)
 *   Hb = S^T S  (bordered: H/2, g/2, loss)        ((nz+1) x (nz+1), ld ldh)
 *   z <- z - step * H^{-1} g                       via Cholesky of Hb
 * host_loss_in = loss(z_in) = sum_k ||L_k^{-1} F_k(z_in)||^2 by TRUE SUBSTITUTION with the factor(s) (one vector; exact to rounding like
 * gpk_gn_loss -- rounds 2-4 returned the squared norm of the F column of the GEMM-only solve instead, ~1e-8 relative error at nugget
 * <= 1e-12 near convergence; since round 6 the structured modes below report the substituted value as well).  WHERE it runs: F(z_in) is
 * written at the start of the call on the handle's stream; from the second call of a handle on, the substitution chain (one single-vector
 * solve per factor + a dot product) is issued on an INTERNAL side stream of the handle (the CU-masked stream of the pipelined phase),
 * next to the end of the step, and joined before the call returns -- the call is still synchronous for the caller and ordered on the
 * handle's stream, but a profiler shows a second stream; the first call of a handle, handles without the two-partition pipeline and
 * gpk_tune(h, 52, 2) run the chain on the handle's stream in front of the solve phase.  host_info = potrf info of H (0 ok).
 * delta (nz,) receives H^{-1} g.  One call = one iteration of the
 * reference's GN_method loop (Hessian, gradient, solve, update, one loss evaluation). */
int gpk_gn_step(gpk_handle h, const gpk_gn_problem* host_prob, double* z, double step_size,
                double* S, int lds, double* Hb, int ldh, double* delta, double* host_loss_in, int* host_info);
/* OPTIONAL structured solve of the elliptic system (not what the reference does per step; off unless W1/W2/v0 are set).  A(z) =
 * [diag(alpha m z^(m-1)); I; 0] has its non-zeros in fixed places and the solve is linear in the right-hand sides, so with
 *   W1 = L^{-1} [I; 0; 0],  W2 = L^{-1} [0; I; 0]   (s_rows x nz each, leading dimension ldw >= nz+1, columns in the internal order of
 *   gpk_gn_step),  v0 = L^{-1} F(0)  (s_rows)
 * computed ONCE by this call (two solves; S is scratch, s_rows x lds), every later gpk_gn_step whose host_prob carries W1, W2, v0, ldw
 * forms  S = [W1 diag(d(z)) + W2 | v0 + W1 (alpha z^m) + W2 z]  in one memory-bound pass instead of the triangular solve.  Same
 * iterates up to rounding (tests/test_gpu_structured.py); the product, the factorisation and the update are unchanged.
 * Systems other than the elliptic one (round 6: GPK_GN_BURGERS, GPK_GN_EIKONAL, GPK_GN_DARCY; needs Dinv / Dinv2 / dinv_block, i.e. the
 * GEMM-only solve path and its leading-zero layout).  Every column of their A(z) has at most ONE entry that depends on z -- Burgers:
 * the PDE row (-alpha v2, -alpha v0, nu: src/PDEs.py:297-299 of the reference), Eikonal: 2 v1 / eps, 2 v2 / eps (:443-445), Darcy: the
 * v3 row of the u-part (f e^{-w0}, -v1, -v2, -w1, -w2: src/InverseProblems.py:131-135) -- so A(z) = A1 diag(d(z)) + A2 with constant
 * 0/1 patterns, and  W1 = L^{-1} A1,  W2 = L^{-1} A2  (all row groups stacked, s_rows x (nz+1) each, ldw even and >= nz+1; v0 is not
 * used and may be NULL) are computed ONCE here.  Every later gpk_gn_step whose host_prob carries W1, W2, ldw forms
 * L^{-1}A(z) = W1 diag(d(z)) + W2 in one memory-bound pass; the column L^{-1}F(z) is still SOLVED every step (one column per factor).
 * Same iterates up to rounding (tests/test_gpu_structured.py: <= 1e-8 relative to the per-step solve, <= 1e-6 to the oracle). */
int gpk_gn_structured_prepare(gpk_handle h, const gpk_gn_problem* host_prob, double* S, int lds, double* W1, double* W2, double* v0, int ldw);
/* OPTIONAL second level of the structured mode (elliptic system): with the Gram blocks of W = [W1 W2],
 *   G = [G11; G12; G21; G22]  (four nz x nz blocks stacked, leading dimension ldg >= nz; Gij = Wi^T Wj),
 *   pvec = [W1^T v0 (nz); W2^T v0 (nz); v0^T v0 (1)],
 * computed ONCE by this call from W1, W2, v0 (host_prob must carry them), gpk_gn_step assembles the bordered matrix directly,
 *   H/2 = D G11 D + D G12 + G21 D + G22,   g/2 = D q1 + q2,   loss = v0^T v0 + a.p1 + z.p2 + a.q1 + z.q2
 *   (D = diag(alpha m z^(m-1)), a = alpha z^m, q1 = G11 a + G12 z + p1, q2 = G21 a + G22 z + p2),
 * in O(nz^2) memory-bound work per step: neither the triangular solve nor the product S^T S is executed; the Cholesky
 * factorisation of H, the solve and the update are unchanged.  Same iterates up to rounding (tests/test_gpu_structured.py).
 * Burgers / Eikonal / Darcy (round 6; host_prob carries the W1, W2 of their structured form, A(z) = A1 diag(d(z)) + A2): the same four
 * blocks; per step  H/2 = D G11 D + D G12 + G21 D + G22  with D = diag(d(z)), and the border from the column w = L^{-1}F(z), which is
 * SOLVED every step (one column per factor):  g/2 = D W1^T w + W2^T w,  loss = w^T w.  pvec (2 nz + 1) is zeroed and not used.
 * H of these systems is ill-conditioned (1e10 .. 1e12) -- the assembled form agrees with S^T S to 1e-14 relative and the steps to
 * 1e-10 (tests/test_gpu_structured.py), inside the 1e-6 parity bound, but it is opt-in like every structured mode. */
int gpk_gn_gram_prepare(gpk_handle h, const gpk_gn_problem* host_prob, double* G, int ldg, double* pvec);
/* Darcy system: the iteration-independent part of the step, computed ONCE per factor (round 6).  The a-part rows of GN_loss,
 * [w1; w2; w0] against L_a (src/InverseProblems.py:137-146 of the reference), do not involve z_old: their block of A(z) is a
 * permutation matrix, so W_a = L_a^{-1} A_a and its contribution H_a = W_a^T W_a to the Gauss-Newton matrix are the same in every
 * step -- the reference recomputes them inside every Hessian_GN because autodiff cannot know.  This call runs exactly the launches a
 * step would (same kernels, same shapes) and keeps their results:
 *   Wa (3 N_d x 3 N_d, ld ldwa >= 3 N_d): the solved a-part block in the step's internal column order; must not alias S;
 *   Ha (3 N_d x 3 N_d, ld ldha >= 3 N_d): the lower triangle of W_a^T W_a;
 * S (s_rows x lds) is scratch.  A gpk_gn_step whose host_prob carries Wa/ldwa/Ha/ldha then skips the a-part's 3 N_d-column solve and
 * its product (an O(N_d^2) add instead); the a-part's F column (which depends on z) is still solved every step.  The iterates are
 * BIT-IDENTICAL to the uncached step (tests/test_gpu_structured.py).  Needs Dinv / Dinv2 (the GEMM-only solve path); on the other
 * paths the fields are ignored.  The caller must drop Wa / Ha when L2 is refactored. */
int gpk_gn_darcy_prepare(gpk_handle h, const gpk_gn_problem* host_prob, double* S, int lds, double* Wa, int ldwa, double* Ha, int ldha);
/* building blocks of gpk_gn_step for the column-sharded multi-GPU step: S <- [A(z) | F(z)] (no solve), y += alpha x */
int gpk_gn_build(gpk_handle h, const gpk_gn_problem* host_prob, const double* z, double* S, int lds);
/* Same with unknown j stored in column n_z-1-j (elliptic system): column c < n_z of [A | F] is then zero above row
 * n_z-1-c -- the layout gpk_gn_step uses internally and gpk_trsm_lz / gpk_gemm_lz exploit.  Round 6: also the Eikonal, Burgers and
 * Darcy systems, in the staircase orders of THEIR gpk_gn_step (unknown groups v1, v2, v0 / the three unknowns of a point interleaved /
 * v1, v2, w1, w2, w0, v0). */
int gpk_gn_build_rev(gpk_handle h, const gpk_gn_problem* p, const double* z, double* S, int lds);
int gpk_axpy(gpk_handle h, int n, double alpha, const double* x, double* y);
/* loss(z) (src/PDEs.py:82-87,278-289,418-430,138-147; src/InverseProblems.py:105-120); work: s_rows doubles. */
int gpk_gn_loss(gpk_handle h, const gpk_gn_problem* host_prob, const double* z, double* work, double* host_loss);
/* Hessian_GN(z,z) and grad_loss(z) as the reference returns them (full symmetric H = 2 A^T Theta^{-1} A, g);
 * H is nz x nz with ld ldh (needs (nz+1) x ldh storage, ldh >= nz+1), g (nz,). */
int gpk_gn_hessian_grad(gpk_handle h, const gpk_gn_problem* host_prob, const double* z,
                        double* S, int lds, double* H, int ldh, double* g);
/* measurement vector(s) F(z) = sol_vec (src/PDEs.py:132-134,338-342,488-497; IP.py:176-186): out (s_rows,) */
int gpk_gn_measurement(gpk_handle h, const gpk_gn_problem* host_prob, const double* z, double* out);

/* Development switches, phase stamps, probes and the micro-benchmarks behind the roofline denominators: include/gpk_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* GPK_H */

// gpk_common.h -- internal declarations shared by the libgpk translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/gpk.h"
#include "../../include/gpk_debug.h"
#ifdef GPK_DEV
#include "../../include/gpk_dev.h"
#endif
#include "../../include/gpk_mg.h"

#define GPK_ERR_ARG (-9001)
#define GPK_ERR_NODEV (-9002)

constexpr int GPK_MAX_TRSV_BLOCKS = 4096;

// Piecewise-linear leading-zero profile ("staircase") of a block of right-hand sides: column c (storage order) is known to be zero
// above row first_row(c).  Segment s covers the columns [c1[s-1], c1[s]) (c1[-1] = 0) with first_row(c) = a[s] + (b[s] - c) / sd[s]
// (b[s] >= every c of the segment; sd huge = a flat step); columns from c1[nseg-1] on are dense.  The elliptic, Eikonal and Burgers
// systems have ONE segment and keep the closed form (lead, lead_div) of gpk_i_gemm; the Darcy system's u-part has three (round 4):
// slope 1 over the v1, v2 columns, flat over w1, w2, slope 1 over w0, v0 (gpk_gn.hip).  nseg = 0: no profile.
struct GpkStair {
    int nseg = 0;
    int c1[4] = {0, 0, 0, 0};
    int a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0}, sd[4] = {1, 1, 1, 1};
};
// smallest first_row over the columns [n0, n1) -- the row from which a tile with those columns has to start its K loop
__host__ __device__ inline int gpk_stair_min(const GpkStair& st, int n0, int n1) {
    int best = 0x7fffffff, lo = 0;
    for (int s = 0; s < st.nseg; ++s) {
        const int hi = st.c1[s];
        const int x0 = n0 > lo ? n0 : lo, x1 = n1 < hi ? n1 : hi;    // overlap of the tile with the segment
        if (x1 > x0) {                                               // non-increasing inside a segment: the right-most column decides
            const int d = st.b[s] - (x1 - 1);
            const int v = st.a[s] + (d > 0 ? d / st.sd[s] : 0);
            best = v < best ? v : best;
        }
        lo = hi;
    }
    if (n1 > lo) best = 0;                                           // dense columns behind the last segment
    return best == 0x7fffffff ? 0 : best;
}
// first column that can be non-zero in the rows [0, rows_end): the smallest c with first_row(c) < rows_end (profiles used by the
// triangular solve are non-increasing in c); ncols if there is none
inline int gpk_stair_first_col(const GpkStair& st, int rows_end, int ncols) {
    int lo = 0, hi = ncols;                                          // invariant: first_row(c) >= rows_end for c < lo, < rows_end for c >= hi
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (gpk_stair_min(st, mid, mid + 1) < rows_end) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Development / tuning switches of ONE handle (round 4: they used to be process-wide globals).  Defaults are the measured choices;
// gpk_tune(handle, key, value) (include/gpk_debug.h) changes one -- tests force every kernel variant through it, A/B measurements
// flip one switch at a time.  The product path never calls gpk_tune.  "key N" = the key of gpk_tune.
struct GpkTune {
    int asm_pairs = 1;                  // key 47: 0 = one column point per lane (8-byte stores) always
    int seq_left_looking = 0;           // key 56: 1 = left-looking rank-64 work inside the 512-blocks of the one-stream Cholesky (experiment, round 6; with key 48 = 2)
    int asm_nt = 0;                     // key 55: 1 = non-temporal 16-byte stores in the two-point evaluator (round 6 A/B, tools/assembly_store_ab.py)
    int dbg = 0;                        // 
    int mt_trsm = 0;                    // key 2: 
    int persistent_ob = 0;              // key 7: 1 = persistent outer-block kernel (slower, see its header)
    int left_looking_panels = 1;        // key 18: 0 = right-looking rank-64 updates also in the pipelined chain
    int panel_mfma = 1;                 // key 21: 0 = first-design panel kernel (potf2_tile: two columns per barrier)
    int panel_unrolled = 1;             // key 41: 0 = the rolled instantiation of the panel kernel
    int fused_panel = 1;                // key 5: 0 = potf2 + trsm launches
    int panel_fused = 1;                // key 48: 0 = rank-64 updates as launches of their own between the panel kernels (round 2)
    int strip = 1;                      // key 3: 0 = 64-row base solves only
    int solve_splitk = 0;               // key 30: largest split-K factor tried for the updates of the inverted-block solve (0 = off)
    int potrf_pipeline_min_n = 2048;    // key 19: plain Cholesky pipelined for orders in [min, max] (max = 0: off, see gpk_i_potrf)
    int potrf_pipeline_max_n = 0;       // key 20: plain Cholesky pipelined for orders in [min, max] (max = 0: off, see gpk_i_potrf)
    int potrf_lookahead = 0;            // key 54 (development build only): 1 = the two-partition plain Cholesky (orders of keys 19 / 20) in its right-looking look-ahead form (round 5: measured, not adopted) instead of the left-looking pipeline
    int potrf_ob = 512;                 // key 51: outer block width of the right-looking factorisation (multiple of 64)
    int pipeline = 1;                   // key 12: 0 = SYRK, then right-looking Cholesky, on one stream
    int pipeline_chain_cus = 32;        // key 13: CUs of the chain partition (rounded to a multiple of 32)
    int pipeline_pre = 1;               // key 17: blocks of the product computed before the fork
    int pipeline_tile = 64;             // key 34: 64 / 128 = tile height of the pipeline's products (parallelism from split-K alone), 0 = automatic (32 rows below 512 tiles).  Phase time at config 2: automatic 3.18-3.23 ms, 64 rows 3.12-3.14, 128 rows 3.47
    int pipeline_lookahead = 0;         // key 26: 0 = block j's update with block j-1 as ONE product after the chain of j-1
    int pipeline_units = 1000;          // key 24: workgroups aimed at per product launch of the pipeline (split-K; 0 = no split).  Measured at config 2, phase time: 0 / 1000 / 1500 / 2000 / 3000 -> 3.85 / 3.57 / 3.58 / 3.60 / 3.62 ms
    int pipeline_max_n = 7000;          // key 14: pipelined only up to this order (with the split-K products, product + factorisation per step: order 6001 7.18 -> 6.50 ms, 7001 13.7 -> 13.6, 8501 22.3 -> 23.3, 10001 35.8 -> 38.6)
    int pipeline_w0 = 512;              // key 28: 
    int pipeline_ob = 512;              // key 29: 
    int probe_chain_cus = 0;            // key 11: overlap probe with a CU-mask partition
    int fused_trsv = 1;                 // key 4: 0 = two launches per block, 1 = fused with data-tagged hand-offs (round 3), 2 = fused with flags (round 1)
    int gemm_extra_lds = 0;             // key 9: bytes of dynamic LDS requested on top (occupancy throttle for overlap experiments)
    int k64_small = 1;                  // key 8: 0 = 64-row tiles only in the K <= 64 kernel
    int band_mb = 192;                  // key 35: MB of A per band of a tall leading-zero launch (0 = no bands).  North-star size, solve phase: no bands 45.2 ms, 48 MB 44.9, 96 MB 43.2, 192 MB 42.6-43.1, 288 MB 43.9, 400 MB 45.4
    int syrk_band = 256;                // key 36: MB of S per band of a large leading-zero SYRK launch (0 = column-major over all rows).  North-star size, the product S^T S: no bands 24.15 ms, 192 MB 23.4, 256 MB 22.9, 384 MB 23.55, 512 MB 24.3
    int big_min = 6000;                 // key 38: launches with at least this many 64 x 64 tiles use the 128 x 128 tile with 16 waves, one workgroup per CU (0 = never).  tools/gemm_big_probe.py, 64 x 64 -> 128 x 64 -> this: NN 10500^3 60.0 / 58.6 / 65.2 TF/s, 8192^3 62.2 / 60.9 / 69.1 on a slow box; north-star solve phase 44.6 -> 43.7 ms with thresholds 3000 .. 6000; at config 2 (threshold 2000) the solve phase loses 5 %
    int big_lower_min = 8000;           // key 50: lower-triangular leading-zero launches (S^T S) with at least this many lower 64 x 64 tiles use the 128 x 128 / 16-wave tile (0 = never)
    int tall_min = 1500;                // key 33: launches with at least this many 64 x 64 tiles use the 128 x 64 / 8-wave tile (0 = never).  Measured (tools/gemm_big_probe.py, 64 x 64 -> 128 x 64): NN 10500^3 64.5 -> 67.9 TF/s, TN 4001^2 x 8400 61.3 -> 66.4, NN 2048 x 16001 x 2048 61.2 -> 65.0, 8192^3 68.6 -> 69.2; in the solve phase at config 2 the 1568-tile update 397 -> 352 us, the 3276-tile one -2 %, the 1260-tile one +10 % (hence the threshold); north-star size: solve 46.0 -> 44.7 ms
    int force_splitk = 0;               // key 25: split K of every eligible gpk_gemm launch into this many chunks (tests)
    int rev_k = 0;                      // key 16: 
    int stagger = 0;                    // key 15: start-time stagger of co-resident GEMM workgroups (experiment)
    int supertile = 0;                  // key 6: 1 = supertile schedule for the leading-zero SYRK (below)
    int sk = 1;                         // key 42: 0 = never, 1 = automatic, 2 = every eligible launch
    int sk_rounds = 6;                  // key 43: automatic mode uses tile lists for launches of fewer than this many rounds of resident workgroups
    int sk_stagger = 2;                 // key 45: start stagger of the co-resident workgroups of a tile-list launch (slot x this x 512 cycles; every workgroup starts at once and has the same amount of work -- without it the four workgroups of a CU run in lock-step, see the kernel)
    int sk_rowclass = 1;                // key 46: 0 = keep the launch's tile order when cutting shares (experiment)
    int row_order = 0;                  // key 49: 1 = row-major, per-XCD-contiguous tile order for the narrow leading-zero products of the pipelined phase (experiment, round 3: fewer re-reads of S by construction, but slower -- sum of the product launches 2.14 -> 2.31 ms, phase 3.20 -> 3.39 ms at config 2, tools/row_order_ab.sh: longest-column-first matters more than the traffic)
    int sk_snap = 4;                    // key 44: a share boundary closer than this many slabs to a tile boundary moves there
    int force_cfg = 0;                  // key 0: development aid (): 0 auto, 1 = 128x128 tiles, 2 = 64x64 tiles
    int use_dinv = 1;                   // key 10: 0 = substitution strips even when the inverses are supplied
    int eikonal_lz = 1;                 // key 23: 0 = dense schedule for the Eikonal, Burgers and Darcy systems; 2 = leading-zero layout with the conservative closed-form staircase for Eikonal (round 2-3) instead of the exact two-segment profile
    int exact_loss = 1;                 // key 52: 1 = gpk_gn_step reports the loss of the iterate it starts from by TRUE SUBSTITUTION (round 5): F(z) is written at the start of the
                                        // step and solved with the factor(s) as a single vector on the GEMM partition's stream next to the END of the step (last panel chain of the
                                        // pipelined phase on the other partition, then the backward solve of the tail on the main stream -- latency chains on an idle chip);
                                        // 2 = the same chain on the main stream in front of the solve phase (+0.43 ms at config 2, what a gpk_gn_loss call costs);
                                        // 0 = the squared norm of the F column of the GEMM-only solve (free, ~1e-8 relative error at nugget <= 1e-12 near convergence; rounds 2-4).
                                        // Measured and lost: the chain next to the SOLVE phase -- on the 32-CU chain partition beside whole-chip GEMM launches the solve grows by
                                        // 0.67 ms, with the first solve launches moved to the 224-CU partition meanwhile by 0.63 ms (the chain alone takes 0.93 ms on 32 CUs)
    int structured = 1;                 // key 40: 0 = ignore W1/W2/v0 (always the triangular solve); 1 = honour W1/W2/v0 only (never the Gram blocks); 2 would be redundant: the Gram level is used whenever G/pvec are set
};

struct gpk_ctx {
    GpkTune tune;                   // development / tuning switches of this handle (gpk_tune)
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int* d_info = nullptr;          // device int: first non-positive pivot (1-based), 0 = none
    double* d_scalars = nullptr;    // small device scratch for reductions (16 doubles)
    double* h_pinned = nullptr;     // 8 doubles of pinned host memory: where the end-of-step scalars (loss, pivot status) land -- a copy into
                                    // pageable memory goes through a staging buffer and a second synchronisation
    int* d_flags = nullptr;         // per-block "solved" epochs of the fused single-vector triangular solve (GPK_MAX_TRSV_BLOCKS ints)
    int trsv_epoch = 0;
    void* d_trsv_gran = nullptr;    // {value, epoch} granules of the fused triangular solve with data-tagged hand-offs (64 per block), allocated on first use
    long long trsv_gran_epoch = 0;
    double* d_loss_work = nullptr;  // F(z) as one contiguous vector for the exact in-step loss (rows doubles, grown on demand)
    size_t loss_work_cap = 0;       // doubles
    // the exact in-step loss runs its chain on the GEMM partition's stream NEXT TO the end of the step (the last panel chain of the pipelined
    // phase on the chain partition -- disjoint CUs -- and the backward single-vector solve of the tail on the main stream: latency chains on an
    // otherwise idle chip): a second granule set for that concurrent solve, and the events main -> side (F(z) is built; side may start) and
    // side -> main (the scalar is there)
    void* d_trsv_gran2 = nullptr;
    long long trsv_gran2_epoch = 0;
    int trsv_alt = 0;
    hipEvent_t ev_loss[2] = {nullptr, nullptr};
    int* d_obflags = nullptr;       // 64 flags of the persistent outer-block Cholesky kernel (epoch-tagged)
    int ob_epoch = 0;
    unsigned panel_loaded = 0;      // running total of "diagonal block loaded" tickets issued to the Cholesky panel kernel (d_flags[GPK_MAX_TRSV_BLOCKS])
    double* d_work = nullptr;       // scratch of the Gauss-Newton step: the out-of-place solution of the Dinv solve
    size_t work_cap = 0;            // bytes
    long work_sig[5] = {0, 0, 0, 0, 0};   // (rows, ld, nrhs, lead, system) of the last solve into d_work: its never-written
                                    // zero region is still valid when the next solve has the same shape
    // SYRK/Cholesky pipeline of the Gauss-Newton step (gpk_i_syrk_potrf): two streams with disjoint CU masks -- the GEMM stream
    // and the "chain" stream that runs the latency-bound panel kernels -- and the events that order them
    hipStream_t pipe_g = nullptr, pipe_c = nullptr;
    int pipe_chain_cus = 0;
    bool pipe_unavailable = false;          // creating the CU-masked streams failed once: this handle stays on the one-stream schedule
    std::vector<hipEvent_t> pipe_ev;        // ordering events (timing disabled)
    std::vector<hipEvent_t> pipe_tev;       // timing events around the SYRK launches (only while prof is on)
    int pipe_tev_used = 0;
    double prof_syrk_ms = 0;                // accumulated duration of the SYRK launches on the GEMM stream (pipelined mode)
    int prof_pipelined = 0;
    // split-K GEMM launches (gpk_gemm.hip, GemmArgs::splitk): requested chunk count for the NEXT gpk_i_gemm calls (0/1 = off; set and
    // cleared by the caller around its launches), the partial-sum workspace and the per-tile arrival counters.  One workspace: only
    // one stream at a time may issue split launches (the GEMM stream of the pipeline does)
    int splitk_req = 0;
    int tile_req = 0;               // 64 / 128: the NEXT gpk_i_gemm calls use 64 x 64 / 128 x 64 tiles whatever the tile count (the caller gets its
                                    // parallelism from split-K); 0 = automatic
    double* d_splitk_ws = nullptr;
    size_t splitk_ws_cap = 0;       // bytes
    unsigned* d_splitk_cnt = nullptr;
    int splitk_cnt_cap = 0;         // counters
    void* sk_cache = nullptr;       // tile-list ("stream-K") plans of the GEMM launches, by launch shape (gpk_gemm.hip)
    int no_sk = 0;                  // != 0: the NEXT gpk_i_gemm calls must not use tile lists (their partial-sum workspace and counters are
                                    // the split-K ones: one stream of a handle at a time -- set around launches on secondary streams)
    double* d_pts = nullptr;        // packed collocation points (SoA), grown on demand
    size_t pts_cap = 0;
    int num_cu = 256;
    GpkStair stair;                 // piecewise staircase of the step being issued (nseg > 0: it replaces the closed form (lead, lead_div) in every
                                    // launch that is given a non-zero `lead`); stair_col0 / stair_row0: global column of the launch's column 0 and
                                    // global row of its k = 0, set by the caller around each gpk_i_gemm call
    int stair_col0 = 0, stair_row0 = 0;
    int stair_base = 0;             // gpk_i_trsm_left_dinv with a piecewise profile: global column of ITS column 0 (a column shard of the sharded step)
    int lead_div = 1;               // slope of the leading-zero staircase while a Gauss-Newton step is being issued: column c of the
                                    // right-hand side is zero above row (lead-1-c) / lead_div (1: elliptic, Eikonal; 3: Burgers)
    hipStream_t side[3] = {nullptr, nullptr, nullptr};   // column-group streams of the multi-RHS triangular solve
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // per-phase HIP-event timing of gpk_gn_step (bench.py roofline): 0 TRSM, 1 SYRK launch, 2 POTRF, 3 TRSV+update
    bool prof = false;
    hipEvent_t pev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double prof_ms[4] = {0, 0, 0, 0};
    int prof_cnt = 0;
    hipEvent_t asm_ev[2] = {nullptr, nullptr};   // around the last gpk_assemble evaluator launch while prof is on (gpk_prof_read_assembly)
    bool asm_timed = false;
    int prof_phase = 0;             // phase the step is issuing right now (last GPK_PROF_MARK index): 0 solve, 1 product + factorisation of Hb, ...
    double prof_flops[4] = {0, 0, 0, 0};   // flops EXECUTED by the GEMM launches of each phase while prof is on (gpk_gemm.hip, prof_count)
    long prof_launches[4] = {0, 0, 0, 0};
    std::string err;
};

int gpk_fail(gpk_handle h, hipError_t e, const char* what, const char* file, int line);
int gpk_bad_arg(gpk_handle h, const char* what);

#define GPK_HIP(h, call)                                                          \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) return gpk_fail((h), e__, #call, __FILE__, __LINE__); \
    } while (0)
#define GPK_TRY(expr)               \
    do {                            \
        int r__ = (expr);           \
        if (r__ != 0) return r__;   \
    } while (0)
#define GPK_LAUNCH_CHECK(h) GPK_HIP((h), hipGetLastError())

// ---- internal (stream-ordered, no host sync) building blocks -------------------------------------------------
// C <- alpha*op(A)*op(B) + beta*C.  lower_only: skip tiles strictly above the diagonal (square C).
int gpk_i_gemm(gpk_handle h, bool ta, bool tb, int m, int n, int k, double alpha, const double* A, int lda,
               const double* B, int ldb, double beta, double* C, int ldc, bool lower_only, int lead = 0, bool tri_a = false,
               bool skip_upper = false);
int gpk_i_potrf(gpk_handle h, double* A, int n, int lda, int pivot_base);               // info -> h->d_info
int gpk_i_potrf_panel(gpk_handle h, double* A, int nrows, int ob, int lda, int pivot_base, bool left_looking = false,
                      void* ev_wait_p1 = nullptr, void* ev_rec_pre = nullptr);   // (hipEvent_t; see gpk_factor.hip)
// Hb <- chol(W^T W) (lower, nc x nc; W is rows x nc with the leading-zero shape `lead` of gpk_i_gemm), the product and the
// factorisation pipelined by 512-column blocks on two CU partitions (gpk_factor.hip); d_loss (device, may be null) receives
// the unfactored last diagonal entry (W^T W)[nc-1][nc-1]
int gpk_i_syrk_potrf(gpk_handle h, const double* W, int ldw, int rows, int nc, int lead, double* Hb, int ldh, double* d_loss);
int gpk_i_trsm_left(gpk_handle h, bool trans, const double* L, int n, int ldl, double* B, int nrhs, int ldb);
// forward solve exploiting leading zeros of the right-hand side columns (see gpk_factor.hip)
int gpk_i_trsm_left_lz(gpk_handle h, const double* L, int n, int ldl, double* B, int nrhs, int ldb, int lead, int row0);
// same, right-hand sides split into independent column groups that run on concurrent streams
int gpk_i_trsm_left_mt(gpk_handle h, bool trans, const double* L, int n, int ldl, double* B, int nrhs, int ldb);
// explicit inverses of the db x db diagonal blocks of L (Dinv: n x db, ld db; db = 256, 512 or 1024) and the all-GEMM forward
// solve built on them: X <- L^{-1} B out of place, B is scratch afterwards (see gpk_factor.hip)
int gpk_i_trtri_diag(gpk_handle h, const double* L, int n, int ldl, double* Dinv, int db);
int gpk_i_trsm_left_dinv(gpk_handle h, const double* L, const double* Dinv, int db, int n, int ldl, double* B, int ldb,
                         double* X, int ldx, int nrhs, int lead, int row0);
int gpk_i_pipe_streams(gpk_handle h);                                                   // the two CU-masked streams of the pipeline (gpk_factor.hip); called by gpk_create
void gpk_i_sk_free(gpk_handle h);                                                      // tile-list plans (gpk_gemm.hip)
int gpk_i_splitk_reserve(gpk_handle h);                                                 // workspace + counters of the split-K launches
int gpk_i_workspace(gpk_handle h, size_t bytes, double** out);                          // handle-owned scratch, grown on demand
int gpk_i_trsm_right_lt(gpk_handle h, const double* L, int n, int ldl, double* X, int m, int ldx);
int gpk_i_trsv(gpk_handle h, bool trans, const double* L, int n, int ldl, double* x);   // x contiguous
int gpk_i_dot(gpk_handle h, const double* x, const double* y, int n, double* d_out);    // d_out device scalar
int gpk_i_ensure_points(gpk_handle h, size_t doubles);
// pieces of the Gauss-Newton step used by the multi-GPU schedule (gpk_gn.hip)
int gpk_i_gn_dims(gpk_handle h, const gpk_gn_problem* p, int* nz, int* rows);
// the column layout gpk_gn_step runs a system in (1 elliptic, 2 Eikonal, 3 Burgers, 4 Darcy, 0 dense) and the handle state that goes with it for
// one call (Eikonal: its two-segment profile in gpk_ctx::stair; Burgers: lead_div = 3); leave resets it.  first_row: the first row of column c
// (storage order, c < nz) that can be non-zero under the layout entered
int gpk_i_gn_layout(gpk_handle h, const gpk_gn_problem* p);
void gpk_i_gn_layout_enter(gpk_handle h, const gpk_gn_problem* p, int rev);
void gpk_i_gn_layout_leave(gpk_handle h);
int gpk_i_gn_first_row(gpk_handle h, int nz, int c);
void gpk_i_gn_darcy_profile(gpk_handle h, int Nd);                   // the Darcy u-part's profile into gpk_ctx::stair (sharded step)
int gpk_i_gn_darcy_add_a(gpk_handle h, const gpk_gn_problem* p, double* Hb, int ldh, int r0, int r1, const double* aF, int ldaf);
int gpk_i_gn_exact_loss(gpk_handle h, const gpk_gn_problem* p, const double* z, double** d_out);   // gpk_tune key 52: loss(z) by substitution on h->stream
int gpk_i_gn_finish(gpk_handle h, const gpk_gn_problem* p, int nz, int rev, const double* Hb, int ldh, double* scratch, double* delta,
                    double* z, double step_size);

static inline int gpk_ceil_div(int a, int b) { return (a + b - 1) / b; }

// gpk_factor.hip -- Cholesky factorisation and triangular solves (fp64), recursive blocking on top of the MFMA GEMM.
//
// Replaces jnp.linalg.cholesky (reference src/PDEs.py:77,273,413; src/InverseProblems.py:102-103) and every
// jnp.linalg.solve with the triangular factor (src/PDEs.py:86,97,143,161,205,288,306,347,429,450,502;
// src/InverseProblems.py:118-119,145-146,190,195).  The reference calls a GENERAL LU solve on the lower-triangular
// L each time; here L is used as what it is.
//
// Structure: all O(n^3) work is delegated to gpk_i_gemm (MFMA); the diagonal blocks are handled by the kernels below.
//   potrf(A)          = two-level right-looking (gpk_i_potrf): 512-column blocks, inside a block one fused panel kernel per 64
//                       columns (potrf_panel_mfma_kernel) with the rank-64 updates riding inside (PanelFuse)
//   syrk_potrf(W)     = chol(W^T W) of the Gauss-Newton step, product and factorisation pipelined by 512-column blocks on two
//                       CU-mask partitions (gpk_i_syrk_potrf / potrf_pipelined)
//   trsm_left(L)      = solve with L11; B2 -= L21 X1; solve with L22 (transposed: mirror image); leaves: 256-row strip kernel
//   trsm_left_dinv(L) = the same recursion with leaves X_k = inv(L_kk) B_k: GEMMs only, for a factor that is used many times
//                       (the inverses of its diagonal BLOCKS come from gpk_i_trtri_diag, by substitution; accuracy: see there)
//   trsm_right_lt     = X1 <- X1 L11^{-T}; X2 -= X1 L21^T; X2 <- X2 L22^{-T}
//   trsv              = single-vector solves in one launch, workgroups chained through data-tagged granules (trsv_gran_kernel)
// Superseded designs (first / third panel kernel, persistent outer-block kernel, flag-chained trsv, potf2 + row-solve launches) were
// removed in round 6 (they had lost every comparison since round 2; the last tree that holds them: git 4be18eb, csrc/dev/gpk_factor_retired.inc).
// True substitution everywhere except trsm_left_dinv.  The recursions split at multiples of 64/128/256 so that sub-blocks stay
// 16-byte aligned for the GEMM's vector loads.
#include "gpk_common.h"

#include <algorithm>
#include <type_traits>

namespace {

constexpr int NB = 64;

__device__ __forceinline__ double bcast(double v, int src_lane) { return __shfl(v, src_lane, 64); }

// broadcast from a COMPILE-TIME lane through the scalar unit (2 x v_readlane_b32): no LDS, no VGPRs, no barrier
__device__ __forceinline__ double bcast_const(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// t[r] = P[r*ld + lane] for r < 64, all loads independent (clamped addresses, no branches): one latency, not 64
__device__ __forceinline__ void load_rows(const double* __restrict__ P, long ld, int nrows, int ncols, int lane, double (&t)[NB]) {
    const int cl = min(lane, ncols - 1);
#pragma unroll
    for (int r = 0; r < NB; ++r) t[r] = P[(long)min(r, nrows - 1) * ld + cl];
}

// ---- 64x64 (or smaller) Cholesky by ONE wave, compact loops ---------------------------------------------------
// The block lives in LDS; it is factored in four panels of 16 columns.  Lane r owns row r: the 16 panel entries of
// its row sit in registers while the panel is factored right-looking (pivot and multipliers are broadcast with
// v_readlane from the lane that owns them), then the trailing columns are updated from LDS in a loop whose body is
// 16 FMAs fed by broadcast ds_read_b128.  Straight-line unrolling of the whole 64x64 factorisation (30+ KB of code
// executed once by one wave) ran 4x slower than this: it is bound by instruction fetch, not by arithmetic.
// development aid: phase time stamps (shader clock) of workgroup 0 (gpk_debug_stamps) -- compiled into the development build only;
// in the product library the `dbg` argument of the kernels is inert
#ifdef GPK_DEV
__device__ unsigned long long gpk_dbg_stamps[16];
#define GPK_STAMP(i) do { if (dbg && blockIdx.x == 0 && threadIdx.x == 0) gpk_dbg_stamps[i] = clock64(); } while (0)
#else
#define GPK_STAMP(i) do { } while (0)
#endif

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int RB = 16;          // register block
constexpr int LB = 32;          // global loads kept in flight per staging round trip (a round trip costs ~1 us)
constexpr int XS = NB + 1;      // odd stride: lane-per-row / lane-per-column accesses hit 64 distinct banks
constexpr int WS = NB + 2;      // even stride: 16-byte aligned broadcast reads of 16 consecutive coefficients

__device__ __forceinline__ double bcast_lane(double v, int src_lane /* wave-uniform */) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// Cross-workgroup data of the persistent kernels travels with agent-scope relaxed atomics (sc1 on gfx950: stores write
// through to memory, loads bypass the XCD-private L2), so a hand-off needs no L2 write-back / invalidate fence.
// Publishing side of such a hand-off: the flag store must not overtake the payload.  A workgroup-scope release fence
// emits NO wait on gfx950 (the payload and the flag may travel to different L2 channels), and the compiler may drop the
// s_waitcnt of an agent-scope fence when it believes the scoreboard is empty (MI355X_MICROARCH.md, "Compiler hazard"),
// so the wait is spelled out: every wave that stored payload executes it before the barrier / the flag store.
__device__ __forceinline__ void gpk_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <bool COH> __device__ __forceinline__ double ld_g(const double* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> __device__ __forceinline__ void st_g(double* p, double v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ---- substitution with a <=64-wide diagonal block: 4 cooperating waves per 64 right-hand sides ------------------------
// TRANS=false: L X = B;  TRANS=true: L^T X = B (index-reversed at load/store so the SAME forward algorithm runs).
// ROWVEC=false: element (j, c) of B at B[j*ldb + c]  (left solve: lanes = consecutive columns, coalesced rows)
// ROWVEC=true : element (j, c) of B at B[c*ldb + j]  (right solve X L^T = A viewed as L X^T = A^T; transposed via LDS)
// Lane = right-hand side in every wave.  Staging is ONE round trip to memory (each wave fetches 16 rows of L and 16 of
// B; a dependent load costs ~3 us here because the operands were just written by another kernel).  The solve walks four
// 16-equation blocks: every wave solves the 16x16 diagonal block redundantly in registers (no hand-off), then the
// remaining equations are updated a quarter per wave (16 FMAs per broadcast ds_read_b128 pair), one barrier per block.
// A single wave doing all of it was LDS-issue bound at ~25 us per launch; ~130 + 63 such launches sit on the critical
// path of every Gauss-Newton step.
struct TrsmShared {
    __attribute__((aligned(16))) double Ws[NB * WS];                 // Ws[a*WS + b]: coefficient of x_b in equation a
    double Xs[NB * XS];                                              // right-hand sides, Xs[a*XS + lane]
    double Ys[NB * XS];                                              // solved values
    double rds[NB];                                                  // 1 / diagonal
};

// L may point to global memory or (fused panel kernel) to the factored diagonal block staged in LDS (ldl = XS then).
// pre != nullptr: the right-hand sides were fetched by the caller (trsm_base_fetch) before L was ready.
template <bool ROWVEC>
__device__ __forceinline__ void trsm_base_fetch(const double* __restrict__ B, long ldb, int nb, int ncols, int blk, double (&tx)[RB]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blk * NB;
    const int lc = min(lane, nb - 1), cc = min(c0 + lane, ncols - 1);
    const int nr = ROWVEC ? min(NB, ncols - c0) : nb;
#pragma unroll
    for (int u = 0; u < RB; ++u) {
        const int r = wave * RB + u;
        tx[u] = ROWVEC ? B[(long)(c0 + min(r, nr - 1)) * ldb + lc] : B[(long)min(r, nb - 1) * ldb + cc];
    }
}

template <bool TRANS, bool ROWVEC, bool COH = false>
__device__ __forceinline__ void trsm_base_body(TrsmShared& sh, const double* __restrict__ L, long ldl, int nb,
                                               double* __restrict__ B, long ldb, int ncols, int blk, int dbg,
                                               const double (*pre)[RB] = nullptr) {
    double* const Ws = sh.Ws; double* const Xs = sh.Xs; double* const Ys = sh.Ys; double* const rds = sh.rds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blk * NB;
    const int c = c0 + lane;
    const int lc = min(lane, nb - 1);
    const int nr = ROWVEC ? min(NB, ncols - c0) : nb;                // rows of the B tile as stored in memory
    const int cc = min(c, ncols - 1);
    // index map: equations are reversed for the transposed solve (only the first nb indices; the rest are padding)
    auto rho = [nb](int i) { return (TRANS && i < nb) ? nb - 1 - i : i; };
    {
        double tl[RB], tx[RB];
#pragma unroll
        for (int u = 0; u < RB; ++u) tl[u] = ld_g<COH>(L + (long)min(wave * RB + u, nb - 1) * ldl + lc);
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = wave * RB + u;
            tx[u] = pre ? (*pre)[u] : (ROWVEC ? B[(long)(c0 + min(r, nr - 1)) * ldb + lc] : B[(long)min(r, nb - 1) * ldb + cc]);
        }
        GPK_STAMP(5);
#pragma unroll
        for (int u = 0; u < RB; ++u) {                               // every (r, lane) pair writes exactly one entry
            const int r = wave * RB + u;
            const bool valid = (r < nb) && (lane < nb);
            const double v = (valid && lane < r) ? tl[u] : 0.0;      // strictly lower part of L, zero elsewhere
            if (TRANS) Ws[rho(lane) * WS + rho(r)] = v;              // L[r][lane]: x_r in equation lane
            else       Ws[r * WS + lane] = v;                        //             x_lane in equation r
            if (lane == r) rds[rho(r)] = valid ? 1.0 / tl[u] : 1.0;
            if (ROWVEC) Xs[rho(lane) * XS + r] = (r < nr && lane < nb) ? tx[u] : 0.0;
            else        Xs[rho(r) * XS + lane] = (r < nb && c < ncols) ? tx[u] : 0.0;
        }
    }
    __syncthreads();
    GPK_STAMP(7);
    const int nblk = (nb + RB - 1) / RB;
    for (int kb = 0; kb < nblk; ++kb) {
        const int r0 = kb * RB;
        double x[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) x[i] = Xs[(r0 + i) * XS + lane];
#pragma unroll
        for (int i = 0; i < RB; ++i) {                               // 16x16 diagonal block (every wave, redundantly)
            const double xi = x[i] * rds[r0 + i];
            x[i] = xi;
#pragma unroll
            for (int j = i + 1; j < RB; ++j) x[j] = fma(-Ws[(r0 + j) * WS + r0 + i], xi, x[j]);
        }
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < RB; ++i) Ys[(r0 + i) * XS + lane] = x[i];
        }
        // remaining equations: groups of 4 rows, dealt round-robin to the waves
        for (int r = r0 + RB + 4 * wave; r < NB; r += 16) {
            const double* __restrict__ wr = Ws + r * WS + r0;
            double acc0 = Xs[r * XS + lane], acc1 = Xs[(r + 1) * XS + lane];
            double acc2 = Xs[(r + 2) * XS + lane], acc3 = Xs[(r + 3) * XS + lane];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                acc0 = fma(-wr[i], x[i], acc0);
                acc1 = fma(-wr[WS + i], x[i], acc1);
                acc2 = fma(-wr[2 * WS + i], x[i], acc2);
                acc3 = fma(-wr[3 * WS + i], x[i], acc3);
            }
            Xs[r * XS + lane] = acc0; Xs[(r + 1) * XS + lane] = acc1;
            Xs[(r + 2) * XS + lane] = acc2; Xs[(r + 3) * XS + lane] = acc3;
        }
        __syncthreads();
    }
    GPK_STAMP(8);
    // equations past the last solved block are padding; solved values leave through Ys, 16 rows per wave
    if (ROWVEC) {
        if (lane < nb) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = wave * RB + u;
                if (r < nr) st_g<COH>(B + (long)(c0 + r) * ldb + lane, Ys[rho(lane) * XS + r]);
            }
        }
    } else {
        if (c < ncols) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = wave * RB + u;
                if (r < nb) B[(long)r * ldb + c] = Ys[rho(r) * XS + lane];
            }
        }
    }
    GPK_STAMP(9);
}

template <bool TRANS, bool ROWVEC>
__global__ __launch_bounds__(256) void trsm_base_kernel(const double* __restrict__ L, long ldl, int nb,
                                                        double* __restrict__ B, long ldb, int ncols, int dbg) {
    GPK_STAMP(4);
    __shared__ TrsmShared sh;
    trsm_base_body<TRANS, ROWVEC>(sh, L, ldl, nb, B, ldb, ncols, blockIdx.x, dbg);
}

// (the first-design panel kernel of round 1 introduced the contract -- the LAST workgroup factors A_jj in place, every other workgroup
// factors its own copy with its 64 rows riding along, load tickets protect the in-place store; it is restated at the kernel below)

// ---- Cholesky panel step, second design (round 2): narrow panels factored inside ONE wave, MFMA trailing updates ------------
// Same contract as potrf_panel_kernel (workgroup 0 factors A_jj in place, workgroup b > 0 factors its own copy and its 64 rows
// below ride along; load tickets protect the in-place store).  What changed is the inside: potf2_tile spends ~720 cycles per
// column (stamps: 46 k cycles for 64 columns with the extra rows) because every pair of columns is an LDS exchange + workgroup
// barrier between an rsqrt chain and a rank-2 update that runs serially behind it.  Here
//   * the 128 x 64 working set (64 rows of A_jj + 64 extra rows) lives in MFMA accumulators for the whole kernel: wave w owns the
//     rows 32w .. 32w+31 (waves 0-1 the diagonal block, waves 2-3 the extra rows), 2 x 4 tiles of 16 x 16;
//   * a panel of PW columns is handed to WAVE 0 through LDS in "lane = row" form -- lane i holds row i of A_jj AND row i of the
//     extra block -- and is factored there right-looking with no LDS traffic and no barrier: the pivot and the multipliers
//     l[c][j] are wave-uniform (v_readlane from the lane that owns row c), the extra rows take the same operations in a second
//     register set;
//   * the factored panel goes back to LDS once and every wave subtracts it from its trailing tiles with v_mfma_f64_16x16x4_f64
//     (K = PW); the tile that holds the NEXT panel is updated and staged for wave 0 right away, the other tiles are updated
//     while wave 0 factors the next panel (between the two barriers of the next round).
// Two workgroup barriers per PW columns instead of one per 2.  Measured (shader-clock stamps, tools/panel_stamp_probe.py, 64 columns
// with extra rows): first design 46 k cycles; PW = 16: 45.6 k (wave 0 alone issues 436 v_readlane + 288 v_fma_f64 per panel, 495
// cycles per column, plus SGPR spills of the multipliers); PW = 8: 38-42 k = per panel ~2050 cycles of factorisation in wave 0
// (259 per column) + ~600 until the slowest wave has finished its deferred updates + ~1950 for the hand-over (LDS operand
// reads -> two dependent MFMAs -> accumulator read-back -> staging writes -> barrier -> wave 0's reads: a chain of LDS and MFMA
// latencies that neither batching the reads nor compile-time tile ranges shortened).  End to end at config 2: product +
// factorisation 3.94 -> 3.83-3.92 ms, Cholesky of Theta 9.35-9.4 -> 9.13-9.16 ms.  The hand-over is what to attack next.
template <int PW>
__device__ __forceinline__ int panel_factor_lane_rows(double* __restrict__ Pc, int p, int l) {
    constexpr int PSW = PW + 2;
    typedef double d2 __attribute__((ext_vector_type(2)));
    double a[PW], x[PW];
#pragma unroll
    for (int i = 0; i < PW / 2; ++i) {
        const d2 t = *reinterpret_cast<const d2*>(Pc + l * PSW + 2 * i);
        a[2 * i] = t.x; a[2 * i + 1] = t.y;
        const d2 u = *reinterpret_cast<const d2*>(Pc + (64 + l) * PSW + 2 * i);
        x[2 * i] = u.x; x[2 * i + 1] = u.y;
    }
    int bad = 0;
    const int cb = __builtin_amdgcn_readfirstlane(PW * p);
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int col = cb + j;
        const double d = bcast_lane(a[j], col);                      // pivot (wave-uniform)
        if (!(d > 0.0) && bad == 0) bad = col + 1;
        const double y0 = __builtin_amdgcn_rsq(d);
        double g = d * y0, hh = 0.5 * y0;
        double e = fma(-g, hh, 0.5);
        g = fma(g, e, g); hh = fma(hh, e, hh);
        e = fma(-g, hh, 0.5);
        g = fma(g, e, g); hh = fma(hh, e, hh);
        const double sq = fma(fma(-g, g, d), hh, g);
        const double rinv = hh + hh;
        a[j] = (l == col) ? sq : a[j] * rinv;
        x[j] = x[j] * rinv;
#pragma unroll
        for (int c = j + 1; c < PW; ++c) {
            const double m = bcast_lane(a[j], cb + c);               // l[cb+c][col]
            a[c] = fma(-a[j], m, a[c]);
            x[c] = fma(-x[j], m, x[c]);
        }
    }
#pragma unroll
    for (int i = 0; i < PW / 2; ++i) {
        *reinterpret_cast<d2*>(Pc + l * PSW + 2 * i) = (d2){a[2 * i], a[2 * i + 1]};
        *reinterpret_cast<d2*>(Pc + (64 + l) * PSW + 2 * i) = (d2){x[2 * i], x[2 * i + 1]};
    }
    return bad;
}

// FUSED (round 3): the kernel carries the rank-64 work that used to sit BETWEEN two panel kernels as launches of their own.
//   * part B, at the start of the panel workgroups: the previous panel (the 64 columns to the left, `prevK` of them) is applied to
//     this panel's columns of my rows while they are loaded -- accumulators -= L[my rows, prev] L[diagonal rows, prev]^T on the
//     matrix cores, operands straight from global memory in fragment form (no LDS);
//   * part A, in EXTRA workgroups behind the panel workgroups (which are dispatched first and all resident): a 64 x 64-tiled
//     product C -= Aop Bop^T that does not depend on this panel and therefore runs NEXT TO its factorisation -- in the
//     left-looking chain of the pipelined phase the contributions of the block's older panels to the NEXT panel's columns
//     (K = 64 .. 384), in the right-looking factorisation the previous panel's rank-64 update of the block's remaining columns.
// The chain per 64 columns becomes panel kernel (+ ~5 us of part B) instead of panel kernel + update launch (10 .. 42 us) + gap.
struct PanelFuse {
    int npanel;                     // workgroups of the panel part (1 + row blocks); blocks beyond run the product
    int prevK;                      // part B: columns of the previous panel (0 or 64), stored directly left of A
    double* gC; const double* gA; const double* gB;   // part A: C (gm x gn) -= gA (gm x gK) gB (gn x gK)^T, all with leading dimension lda
    int gm, gn, gK, gtn;            // gtn = column tiles of C
};

typedef double d4u __attribute__((ext_vector_type(4), aligned(8)));

// extra workgroups of the fused panel kernel: one 64 x 64 tile of C -= A B^T, K a multiple of 16, operands in fragment form straight
// from global memory (lane (li, lk) of a 16-row tile loads the 4 consecutive k's 4 lk .. 4 lk + 3 of row li: MFMA k-slot (step t, lane
// group lk) then stands for k = 4 lk + t in BOTH operands, which is all a contraction needs), one 16-deep chunk prefetched ahead.
// Meant for SHORT products (the rank-64 updates of the right-looking factorisation): fragment-shaped global loads keep the address
// units busy twice as long as full-line loads staged through LDS, and with K = 64 .. 384 on the 32-CU chain partition this product
// needed 48 - 96 us next to a 22 us panel (32 x 64 tiles, four chunks in flight: no better) -- see gpk_i_potrf_panel.
constexpr int PFM = 64;                                              // tile rows of the fused product
__device__ __forceinline__ void panel_fused_product(const PanelFuse& f, long lda, int tile) {
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int li = l & 15, lk = l >> 4;
    const int tm = tile / f.gtn, tn = tile - tm * f.gtn;
    const int m0 = tm * 64 + 32 * (w >> 1), n0 = tn * 64 + 32 * (w & 1);
    const double* __restrict__ pa[2]; const double* __restrict__ pb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        pa[i] = f.gA + (long)min(m0 + 16 * i + li, f.gm - 1) * lda + 4 * lk;
        pb[i] = f.gB + (long)min(n0 + 16 * i + li, f.gn - 1) * lda + 4 * lk;
    }
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    d4u a0[2], b0[2], a1[2], b1[2];
    auto ld = [&](int kc, d4u (&a)[2], d4u (&b)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { a[i] = *reinterpret_cast<const d4u*>(pa[i] + kc); b[i] = *reinterpret_cast<const d4u*>(pb[i] + kc); }
    };
    auto mm = [&](const d4u (&a)[2], const d4u (&b)[2]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    };
    const int nch = f.gK / 16;
    ld(0, a0, b0);
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) ld(16 * (c + 1), a1, b1);
        mm(a0, b0);
        if (c + 1 >= nch) break;
        if (c + 2 < nch) ld(16 * (c + 2), a0, b0);
        mm(a1, b1);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * i + lk + 4 * r;
            if (row >= f.gm) continue;
            double* __restrict__ crow = f.gC + (long)row * lda;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + 16 * j + li;
                if (col < f.gn) crow[col] -= acc[i][j][r];
            }
        }
}

template <int PW, bool UNROLLED, bool FUSED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void potrf_panel_mfma_kernel(double* __restrict__ A, long lda, int nb, int below,
                                                               int* info, int pivot_base, unsigned* loaded, unsigned target, int dbg, PanelFuse fuse) {
    static_assert(PW == 8 || PW == 16, "panel width");
    if (FUSED && (int)blockIdx.x >= fuse.npanel) {                   // part A: independent of this panel, next to its factorisation
        panel_fused_product(fuse, lda, (int)blockIdx.x - fuse.npanel);
        return;
    }
    const int ngrid = FUSED ? fuse.npanel : (int)gridDim.x;          // workgroups of the panel part
    constexpr int PSW = PW + 2;                                      // row stride of a panel buffer: 16-byte aligned rows, conflict-free b64 reads
    constexpr int KS = PW / 4;                                       // MFMA k-steps per panel
    constexpr int HPT = 16 / PW;                                     // panels per 16-column tile
    __shared__ __attribute__((aligned(16))) double Pb[2][128 * PSW];
    __shared__ int sh_bad, sh_expired;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = l & 15, lk = l >> 4;
    // (the LAST workgroup stores A_jj: it waits for tickets of workgroups with smaller indices only, which are dispatched first --
    // with workgroup 0 in that role a grid that is not fully resident can starve, see potrf_panel_la_kernel)
    const bool tall = (int)blockIdx.x + 1 < ngrid;
    const int c0 = (int)blockIdx.x * NB;
    const int xrows = tall ? min(NB, below - c0) : 0;
    double* __restrict__ Xg = A + (long)(nb + (tall ? c0 : 0)) * lda;   // my 64 rows below the diagonal block
    const bool rows_diag = (w < 2);                                  // waves 0, 1: rows of A_jj; waves 2, 3: extra rows
    const bool active = rows_diag || tall;
#ifdef GPK_DEV
#define PANEL_STAMP(i) do { if (dbg && blockIdx.x == 1 && threadIdx.x == 0) gpk_dbg_stamps[i] = clock64(); } while (0)
#else
#define PANEL_STAMP(i) do { } while (0)
#endif
    PANEL_STAMP(0);
    if (tid == 0) sh_bad = 0;

    // ---- load into accumulator layout: tile (rt, ct) lane l reg r = row 32w + 16rt + lk + 4r, column 16ct + li.  Branch-free
    // (clamped addresses, all 32 loads in flight at once; with the loads inside if/else arms the compiler waited for each one)
    d4 acc[2][4];
    {
        const double* __restrict__ base = (rows_diag || !tall) ? A : Xg;     // (waves 2, 3 of workgroup 0 read A and discard it)
        const int rmax = (rows_diag || !tall) ? nb - 1 : xrows - 1;
        const int rlim = rows_diag ? nb : xrows;                              // rows >= rlim are padding (workgroup 0, waves 2-3: xrows = 0)
        const int rbase = 32 * (w & 1) + lk;                                  // row inside my 64-row block (A_jj or extra rows)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rbase + 16 * rt + 4 * r, col = 16 * ct + li;
                    acc[rt][ct][r] = base[(long)min(row, rmax) * lda + min(col, nb - 1)];
                }
        if (FUSED && fuse.prevK > 0) {
            // part B: acc -= L[my rows, prev] L[rows of A_jj, prev]^T, the previous panel's columns sit directly left of A (and of Xg).
            // Fragment loads as in panel_fused_product (k-slot (t, lk) = column 4 lk + t of a 16-column chunk); rows clamped like the
            // loads above (clamped rows only feed accumulator entries that the padding fix-up below overwrites).
            const double* __restrict__ pa[2]; const double* __restrict__ pbq[4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) pa[rt] = base + (long)min(32 * (w & 1) + 16 * rt + li, rmax) * lda - fuse.prevK + 4 * lk;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) pbq[ct] = A + (long)min(16 * ct + li, nb - 1) * lda - fuse.prevK + 4 * lk;
#pragma unroll 1
            for (int kc = 0; kc < fuse.prevK; kc += 16) {
                d4u a[2], b[4];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) a[rt] = *reinterpret_cast<const d4u*>(pa[rt] + kc);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) b[ct] = *reinterpret_cast<const d4u*>(pbq[ct] + kc);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[rt][t], b[ct][t], acc[rt][ct], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rbase + 16 * rt + 4 * r, col = 16 * ct + li;
                    const double pad = (rows_diag && row == col) ? 1.0 : 0.0;    // identity padding of A_jj, zero padding of the extra rows
                    acc[rt][ct][r] = (row < rlim && col < nb) ? acc[rt][ct][r] : pad;
                }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // my loads of A_jj have landed
    // my two tiles of one column tile -> the PW columns of panel q in P[row][PW] (lanes of the other half of the tile sit out)
    auto stage = [&](double* __restrict__ P, int q, const d4 (&t0), const d4 (&t1)) {
        if (HPT == 1 || (li / PW) == (q % HPT)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                P[(32 * w + lk + 4 * r) * PSW + (li % PW)] = t0[r];
                P[(32 * w + 16 + lk + 4 * r) * PSW + (li % PW)] = t1[r];
            }
        }
    };
    // Trailing update of my tiles with the panel in P: acc[rt][ct] -= P[my rows of tile rt] * P[rows 16ct .. 16ct+15]^T for the column
    // tiles LO..HI (compile-time: with run-time bounds the compiler turned the tile loop into branchy code with one LDS wait per
    // tile, 2000+ cycles for four MFMAs).  All LDS operands are requested first (one latency), then the MFMAs run back to back.
    // Tiles of A_jj strictly above the diagonal are never used and skipped (wave-uniform tests).
    auto update_static = [&](const double* __restrict__ P, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        double a0[KS], a1[KS], bf[4][KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            a0[ks] = -P[(32 * w + li) * PSW + 4 * ks + lk];
            a1[ks] = -P[(32 * w + 16 + li) * PSW + 4 * ks + lk];
        }
#pragma unroll
        for (int ct = LO; ct <= HI; ++ct)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bf[ct][ks] = P[(16 * ct + li) * PSW + 4 * ks + lk];   // (rows < 64 always exist in the buffer)
#pragma unroll
        for (int ct = LO; ct <= HI; ++ct) {
            if (16 * ct < nb) {
                if (!rows_diag || 2 * w >= ct) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) acc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks], bf[ct][ks], acc[0][ct], 0, 0, 0);
                }
                if (!rows_diag || 2 * w + 1 >= ct) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) acc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ks], bf[ct][ks], acc[1][ct], 0, 0, 0);
                }
            }
        }
    };
    auto update_from = [&](const double* __restrict__ P, int lo, bool to_end) {     // tiles lo..(to_end ? 3 : lo), dispatched to static code
        using std::integral_constant;
        if (to_end) {
            switch (lo) {
                case 0: update_static(P, integral_constant<int, 0>{}, integral_constant<int, 3>{}); break;
                case 1: update_static(P, integral_constant<int, 1>{}, integral_constant<int, 3>{}); break;
                case 2: update_static(P, integral_constant<int, 2>{}, integral_constant<int, 3>{}); break;
                case 3: update_static(P, integral_constant<int, 3>{}, integral_constant<int, 3>{}); break;
                default: break;
            }
        } else {
            switch (lo) {
                case 0: update_static(P, integral_constant<int, 0>{}, integral_constant<int, 0>{}); break;
                case 1: update_static(P, integral_constant<int, 1>{}, integral_constant<int, 1>{}); break;
                case 2: update_static(P, integral_constant<int, 2>{}, integral_constant<int, 2>{}); break;
                case 3: update_static(P, integral_constant<int, 3>{}, integral_constant<int, 3>{}); break;
                default: break;
            }
        }
    };
    if (active) stage(Pb[0], 0, acc[0][0], acc[1][0]);
    __syncthreads();
    if (tall && tid == 0) __hip_atomic_fetch_add(loaded, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ticket: A_jj has been read
    PANEL_STAMP(1);

    const int npan = (nb + PW - 1) / PW;
    // (a ROLLED loop over the panels: the factorisation code of wave 0 exists once; every accumulator index below is static, the
    // panel index only appears in wave-uniform tests)
    auto iter = [&](const int p) __attribute__((always_inline)) {
        double* __restrict__ Pc = Pb[p & 1];                         // panel p (staged; factored between the barriers)
        double* __restrict__ Pq = Pb[(p + 1) & 1];                   // panel p-1 (still needed for the deferred updates), then panel p+1
        if (p > 0) __syncthreads();                                  // panel p staged by every wave
        if (p == 2) PANEL_STAMP(3);
        if (w == 0) {
            const int bad = panel_factor_lane_rows<PW>(Pc, p, l);
            if (p == 2) PANEL_STAMP(4);
            if (bad && l == 0 && sh_bad == 0) sh_bad = bad;          // (only wave 0 writes; panels in order -> the first one sticks)
        } else if (active && p > 0) {
            // deferred part of panel p-1: every tile to the right of the one that holds panel p (that one was updated at the end of
            // the previous round); runs while wave 0 factors
            update_from(Pq, p / HPT + 1, true);
        }
        __syncthreads();                                             // factored panel (L rows 0..63, X rows 64..127) in Pc
        if (p == 2) PANEL_STAMP(5);
        if (p == 3) PANEL_STAMP(7);
        if (active) {
            const int tp = p / HPT;                                  // tile that holds panel p
            const int tn = (p + 1) / HPT;                            // tile that holds panel p + 1
            if (p + 1 < npan) {
                // the tile with the next panel, now; wave 0 (which has no deferred phase) brings its other tiles up to date as well
                update_from(Pc, tn, w == 0);
            }
            // final values of panel p into my tiles (after the update: with two panels per tile the update above also touched them)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                if (ct == tp && (HPT == 1 || (li / PW) == (p % HPT))) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][ct][r] = Pc[(32 * w + lk + 4 * r) * PSW + (li % PW)];
                        acc[1][ct][r] = Pc[(32 * w + 16 + lk + 4 * r) * PSW + (li % PW)];
                    }
                }
            }
            if (p + 1 < npan) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    if (ct == tn) stage(Pq, p + 1, acc[0][ct], acc[1][ct]);
            }
        }
        if (p == 2) PANEL_STAMP(6);
    };
    // ROLLED: the code exists once, but the tile that holds panel p+1 is selected at run time, which the compiler implements by moving
    // whole accumulator sets between registers (dozens of v_accvgpr_mov per panel).  UNROLLED: every tile index is a constant.  And
    // left to itself the register allocator spread the unrolled kernel over 201 VGPRs + 144 AGPRs (one workgroup per CU); told to fit
    // three waves per SIMD (amdgpu_waves_per_eu) it needs 138 VGPRs, no AGPRs, no scratch -- and is faster still.  Cycles per 64 columns
    // (tools/panel_stamp_probe.py): rolled 42.8 k, rolled with the occupancy hint 33.7 k, unrolled 28.3 k, unrolled with the hint 24.4 k.
    if (UNROLLED) {
#pragma unroll
        for (int p = 0; p < 64 / PW; ++p) {
            if (p >= npan) break;
            iter(p);
        }
    } else {
#pragma unroll 1
        for (int p = 0; p < npan; ++p) iter(p);
    }
    PANEL_STAMP(2);

    // ---- store: workgroup 0 the lower triangle of A_jj (after everybody has read it), the others their extra rows
    if (!tall) {
        __syncthreads();
        if (tid == 0) {
            int it = 0, ex = 0;
            while ((int)(__hip_atomic_load(loaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++it > (1 << 24)) { ex = 1; break; }
            }
            sh_expired = ex;
        }
        __syncthreads();
        if (sh_expired) {
            if (tid == 0) atomicCAS(info, 0, -1);
            return;
        }
        if (rows_diag) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 32 * w + 16 * rt + lk + 4 * r, col = 16 * ct + li;
                        if (row < nb && col <= row) A[(long)row * lda + col] = acc[rt][ct][r];
                    }
        }
        const int bad = sh_bad;
        if (bad && bad <= nb && tid == 0) atomicCAS(info, 0, pivot_base + bad);
    } else if (!rows_diag) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int xr = 32 * (w - 2) + 16 * rt + lk + 4 * r, col = 16 * ct + li;
                    if (xr < xrows && col < nb) Xg[(long)xr * lda + col] = acc[rt][ct][r];
                }
    }
}

// ---- forward substitution with a <=256-wide diagonal block, fused: one launch per 256-row strip ------------------------
// L X = B for a diagonal block of up to SB = 256 rows; one workgroup (8 waves) owns SNC = 16 right-hand-side columns and
// keeps its 256 x 16 slice of B in LDS for the whole solve.  The strip is walked in blocks of 8 equations:
//   solve   : 8 x 8 triangular block by true substitution, lane = column; the 28 coefficients are wave-uniform and come
//             through the scalar cache into SGPRs (all of them fit, so the loads are issued once, ahead of the chain);
//   update  : the 16-row tiles below get  X_t -= L[t][b] X_b  on the matrix cores (v_mfma_f64_16x16x4_f64, K = 8); the
//             wave that updates the tile holding block b+1 solves block b+1 right away and does nothing else in that
//             step, the other seven waves share the tiles further down, so their updates overlap the substitution.  When
//             block b+1 is the lower half of block b's own tile, that tile gets a half update (operand rows 0..7 zeroed).
//   one barrier per step; the L tiles (16 x 8, one coalesced 16-byte load per lane) are fetched three steps ahead into
//   registers and pass through a wave-private LDS tile to reach the MFMA operand layout.
// Measured (tools/strip_probe.py, n = 256, 4001 columns): 37 us per launch = ~18 us skeleton (launch, staging round trip,
// 31 barriers, store) + 5 us tile requests + 7 us substitutions + 8 us MFMA updates.  What the round-1 ablations found:
//   * operand-layout loads straight from memory = 64 line accesses per instruction (3000 cycles per step) -> coalesced
//     16-byte loads + a wave-private LDS transposition;
//   * coefficients through s_load: scalar loads share lgkmcnt with LDS and return out of order, so every LDS wait of
//     the chain became a scalar-memory wait -> 8 x 8 diagonal blocks staged in LDS once;
//   * rotating prefetch registers with moves makes each step wait for the youngest load -> three sets used in rotation
//     by unrolling, loads issued through inline asm with an explicit s_waitcnt vmcnt(6);
//   * 32 columns per workgroup: the MFMA pipe of the one CU is the bound (126 workgroups) -> 16 columns (251 workgroups);
//   * 512-row strips: 96 us, no better than two 256-row strips and the K = 256 update between them.
// No inverse of a diagonal block is ever formed.
constexpr int SB = 256, SNC = 16, SNT = 512;
constexpr int NTL = (SB / 16 - 1 + 6) / 7;                            // row tiles per wave and step (the owner takes one, seven waves share the rest)
constexpr int SCT = SNC / 16;                                        // 16-column MFMA tiles per workgroup
constexpr int SXS = (SNC % 32 == 16) ? SNC : SNC + 16;               // LDS row stride: rows k, k+1 must be 128 B apart mod 256
constexpr int SRPP = SNT / SNC;                                      // rows staged per pass
constexpr int SAS = 10;                                              // L tile (16 x 8) row stride: operand reads conflict-free

__global__ __launch_bounds__(SNT) void trsm_strip_kernel(const double* __restrict__ L, long ldl, int n,
                                                         double* __restrict__ B, long ldb, int ncols, int dbg) {
    GPK_STAMP(10);
    typedef double d2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) double Xs[SB * SXS];
    __shared__ double rd[SB];                                        // 1 / diagonal
    __shared__ __attribute__((aligned(16))) double Ld[SB * 8];       // the 8x8 diagonal blocks, Ld[64 blk + 8 j + i] = L[8 blk + j][8 blk + i]
    __shared__ __attribute__((aligned(16))) double At[(SNT / 64) * NTL * 16 * SAS];   // per wave: NTL 16x8 tiles of L
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: role tests become scalar branches
    const int li = lane & 15, lk = lane >> 4;
    const int c0 = blockIdx.x * SNC;
    const int nblocks = n >> 3, ntiles = n >> 4;                     // the host guarantees n % 16 == 0

    // Roles in update step b (X_b solved): tn = tile of block b+1; wave (tn mod 8) owns it; wave at distance j >= 1 from the
    // owner takes tiles tn + j, tn + j + 7, ... (NTL of them; >= ntiles: none).
    auto first_tile = [wave](int b) {
        const int tn = (b + 1) >> 1;
        const int j = (wave - tn) & 7;
        return tn + j;                                               // j == 0: the owner's single tile
    };
    auto load_a = [&](int b, d2 (&a)[NTL]) {                         // clamped, branch-free
        const int bc = min(b, nblocks - 1);
        const int t0 = first_tile(bc);
#pragma unroll
        for (int k = 0; k < NTL; ++k) {
            const int t = min(t0 + 7 * k, ntiles - 1);
            const double* p = L + (long)(16 * t + (lane >> 2)) * ldl + 8 * bc + 2 * (lane & 3);
            // issued through inline asm so that the s_waitcnt is OURS (vmcnt(6) before use: the two younger sets stay in
            // flight); the compiler's own insertion fell back to vmcnt(0) across the loop's control flow
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[k]) : "v"(p) : "memory");
        }
    };
    static_assert(NTL == 3, "wait_set below names the registers of one set explicitly");
    auto wait_set = [&](d2 (&a)[NTL]) {                              // the oldest set of the three outstanding ones has landed
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) :: "memory");
    };
    double* const at = At + wave * (NTL * 16 * SAS);
    d2 a0[NTL], a1[NTL], a2[NTL];
    {   // stage the slice of B (columns >= ncols as zeros) and the reciprocal diagonal: one round trip to memory
        const int col = tid % SNC, rb = tid / SNC;
        const int cc = min(c0 + col, ncols - 1);
        double v[SB / SRPP];
#pragma unroll
        for (int i = 0; i < SB / SRPP; ++i) v[i] = B[(long)min(rb + SRPP * i, n - 1) * ldb + cc];
        // row (tid) of its 8x8 diagonal block: 64 contiguous bytes per thread.  (Coefficients as LDS broadcasts, not
        // scalar loads: s_load shares the lgkmcnt counter with LDS and returns out of order, so with scalar loads in
        // flight every LDS wait of the chain became a wait for scalar-memory latency.)
        d2 dr[4];
        const int rr = min(tid, n - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) dr[i] = *reinterpret_cast<const d2*>(L + (long)rr * ldl + (rr & ~7) + 2 * i);
#pragma unroll
        for (int i = 0; i < SB / SRPP; ++i) Xs[(rb + SRPP * i) * SXS + col] = (rb + SRPP * i < n && c0 + col < ncols) ? v[i] : 0.0;
        if (tid < SB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<d2*>(Ld + 8 * tid + 2 * i) = dr[i];
            const int jj = tid & 7;
            double dg = dr[0].x;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i == jj) dg = (i & 1) ? dr[i >> 1].y : dr[i >> 1].x;
            rd[tid] = 1.0 / dg;
        }
    }
    __syncthreads();
    // (the compiler has waited for every load of its own above: from here on the only vector-memory operations in flight
    // are the tile requests below, three per step, consumed in order)
    load_a(0, a0); load_a(1, a1); load_a(2, a2);
    GPK_STAMP(11);

    auto solve = [&](int b) {                                        // lanes 32..63 mirror lanes 0..31
        const int col = lane % SNC;
        double* __restrict__ xp = Xs + 8 * b * SXS + col;
        const double* __restrict__ Ls = Ld + 64 * b;
        double x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = xp[j * SXS];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double e = x[j], o = 0.0;
#pragma unroll
            for (int i = 0; i < j; ++i) {
                if (i & 1) o = fma(-Ls[8 * j + i], x[i], o);
                else       e = fma(-Ls[8 * j + i], x[i], e);
            }
            x[j] = (e + o) * rd[8 * b + j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xp[j * SXS] = x[j];
    };
    // tile t -= L[t][b] * X_b; `half`: only rows 8..15 of the tile (the tile is block b's own)
    auto update = [&](int t, int slot, const double (&nb)[SCT][2], bool half) {
        double a[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const double v = at[slot * 16 * SAS + li * SAS + 4 * q + lk];
            a[q] = (half && li < 8) ? 0.0 : v;
        }
        double* __restrict__ xt = Xs + (16 * t + lk) * SXS + li;
        d4 acc[SCT];
#pragma unroll
        for (int c = 0; c < SCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][r] = xt[4 * r * SXS + 16 * c];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < SCT; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], nb[c][q], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < SCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) xt[4 * r * SXS + 16 * c] = acc[c][r];
    };

    // One step.  `au` holds this step's L tiles (requested three steps ago); they go to the wave-private LDS tile first and
    // the same registers immediately take the request for step b+3 -- three register sets used in rotation by unrolling,
    // NOT by moving values between sets (a move would have to wait for the youngest load: that alone made every step
    // cost a full memory round trip, ~2900 cycles).
    auto step = [&](int b, d2 (&au)[NTL]) {
        wait_set(au);
#pragma unroll
        for (int k = 0; k < NTL; ++k)
            *reinterpret_cast<d2*>(at + k * 16 * SAS + (lane >> 2) * SAS + 2 * (lane & 3)) = au[k];
        load_a(b + 3, au);
        const int tn = (b + 1) >> 1;                                 // tile of the next block
        const int t0 = first_tile(b);
        if (t0 < ntiles) {
            double nb[SCT][2];                                       // -X_b in MFMA B-operand layout: B[k = 4q + lk][n = li]
            const double* __restrict__ xs = Xs + (8 * b + lk) * SXS + li;
#pragma unroll
            for (int c = 0; c < SCT; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) nb[c][q] = -xs[4 * q * SXS + 16 * c];
            update(t0, 0, nb, t0 == (b >> 1));
            if (t0 == tn) {
                solve(b + 1);                                        // this wave owns the next diagonal block
            } else {
#pragma unroll
                for (int k = 1; k < NTL; ++k)
                    if (t0 + 7 * k < ntiles) update(t0 + 7 * k, k, nb, false);
            }
        }
        __syncthreads();
    };
    if (wave == 0) solve(0);
    __syncthreads();
    GPK_STAMP(12);
    for (int b = 0; b < nblocks - 1; b += 3) {
        step(b, a0);
        if (b == 0) GPK_STAMP(13);
        if (b + 1 >= nblocks - 1) break;
        step(b + 1, a1);
        if (b + 2 >= nblocks - 1) break;
        step(b + 2, a2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // drain the clamped tail requests
    GPK_STAMP(14);
    {
        const int col = tid % SNC, rb = tid / SNC;
        if (c0 + col < ncols) {
#pragma unroll
            for (int i = 0; i < SB / SRPP; ++i)
                if (rb + SRPP * i < n) B[(long)(rb + SRPP * i) * ldb + c0 + col] = Xs[(rb + SRPP * i) * SXS + col];
        }
    }
    GPK_STAMP(15);
}

// ---- single-vector triangular solve: 64-wide diagonal block by one wave (lane = equation) ---------------------
template <bool TRANS>
__global__ __launch_bounds__(64) void trsv_diag_kernel(const double* __restrict__ L, long ldl, int nb, double* __restrict__ x) {
    __shared__ double T[NB * (NB + 1)];
    const int lane = threadIdx.x;
    {
        double t[NB];
        load_rows(L, ldl, nb, nb, lane, t);
#pragma unroll
        for (int r = 0; r < NB; ++r) T[r * (NB + 1) + lane] = (r < nb && lane < nb) ? t[r] : ((r == lane) ? 1.0 : 0.0);
    }
    __syncthreads();
    double b = (lane < nb) ? x[lane] : 0.0;
    const double rd = 1.0 / T[lane * (NB + 1) + lane];
    double res = 0.0;
    if (!TRANS) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const double xj = bcast(b * rd, j);
            if (lane == j) res = xj;
            b -= T[lane * (NB + 1) + j] * xj;                         // L[lane][j]; only lanes > j matter
        }
    } else {
#pragma unroll
        for (int j = NB - 1; j >= 0; --j) {
            const double xj = bcast(b * rd, j);
            if (lane == j) res = xj;
            b -= T[j * (NB + 1) + lane] * xj;                         // L[j][lane]; only lanes < j matter
        }
    }
    if (lane < nb) x[lane] = res;
}


// ---- the same solve with DATA-TAGGED hand-offs (round 3) -------------------------------------------------------------------
// In trsv_fused_kernel a link of the chain costs 4.1 us: x_w goes out as 64 write-through stores, the wave waits for their
// acknowledgement, then stores the flag; the next workgroup polls the flag, passes a barrier, and every thread fetches 16 x values
// with cache-bypassing loads.  Here x travels as 16-byte granules {value, epoch} written by ONE write-through store per lane and
// polled directly (MI355X_MICROARCH.md, "handoff-1to1": a granule is either old or complete, no separate flag, no drain): wave 0 of
// the consumer loads the 64 granules of block d (one 16-byte sc1 load per lane) until every tag carries this launch's epoch, drops
// the values into LDS (double-buffered: one barrier per link), and all threads take their 16 x values from there.  The plain x
// vector is written as well, off the critical path.  Granule buffer: 64 per block, tags never need clearing (the epoch grows).
struct TrsvGran { double x; long long tag; };

template <bool TRANS>
__global__ __launch_bounds__(256) void trsv_gran_kernel(const double* __restrict__ L, long ldl, int n, double* x,
                                                        TrsvGran* gran, long long epoch) {
    __shared__ double T[NB * (NB + 1)];
    __shared__ double part[4][NB];
    __shared__ double xs[2][NB];
    __shared__ int sh_dead;
    typedef double g2 __attribute__((ext_vector_type(2)));
    const int nblk = (n + NB - 1) / NB;
    const int w = blockIdx.x;
    const int bid = TRANS ? nblk - 1 - w : w;
    const int r0 = bid * NB, nb = min(NB, n - r0);
    const int tid = threadIdx.x, r = tid & 63, q = tid >> 6;         // thread = equation r, columns 16q .. 16q+15 of a tile
    const int rc = min(r, nb - 1);
    if (tid == 0) sh_dead = 0;
    auto load_tile = [&](int d, double (&t)[16]) {
        const int dbid = TRANS ? nblk - 1 - d : d;
        const int c0 = dbid * NB, cnb = min(NB, n - c0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = min(q * 16 + i, cnb - 1);
            t[i] = TRANS ? L[(long)(c0 + c) * ldl + r0 + rc] : L[(long)(r0 + rc) * ldl + c0 + c];
        }
    };
    double tcur[16], tnext[16];
    if (w > 0) load_tile(0, tcur);
    {   // diagonal block -> LDS, identity-padded
        double dv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) dv[u] = L[(long)(r0 + min(q * 16 + u, nb - 1)) * ldl + r0 + rc];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int row = q * 16 + u;
            T[row * (NB + 1) + r] = (row < nb && r < nb) ? dv[u] : ((row == r) ? 1.0 : 0.0);
        }
    }
    double acc = 0.0;
    for (int d = 0; d < w; ++d) {
        if (d + 1 < w) load_tile(d + 1, tnext);
        const int dbid = TRANS ? nblk - 1 - d : d;
        if (q == 0) {                                                // wave 0: poll the 64 granules of block d, one per lane
            const TrsvGran* gp = gran + (long)dbid * NB + r;
            g2 v;
            int it = 0;
            for (;;) {
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(gp) : "memory");
                const long long tag = __double_as_longlong(v.y);
                if (__all(tag == epoch)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++it > (1 << 22)) { if (r == 0) sh_dead = 1; break; }
            }
            xs[d & 1][r] = v.x;                                      // (rows past the end of a short last block carry x = 0)
        }
        __syncthreads();
        const double* __restrict__ xv = xs[d & 1] + q * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = fma(tcur[i], xv[i], acc);
#pragma unroll
        for (int i = 0; i < 16; ++i) tcur[i] = tnext[i];
    }
    part[q][r] = acc;
    __syncthreads();
    if (q == 0) {                                                    // wave 0: substitution on the diagonal block
        double b = (r < nb) ? x[r0 + r] - ((part[0][r] + part[1][r]) + (part[2][r] + part[3][r])) : 0.0;
        const double rd = 1.0 / T[r * (NB + 1) + r];
        double res = 0.0;
        if (!TRANS) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const double xj = bcast_lane(b * rd, j);
                if (r == j) res = xj;
                b = fma(-T[r * (NB + 1) + j], xj, b);
            }
        } else {
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                const double xj = bcast_lane(b * rd, j);
                if (r == j) res = xj;
                b = fma(-T[j * (NB + 1) + r], xj, b);
            }
        }
        if (sh_dead) res = __builtin_nan("");
        if (r >= nb) res = 0.0;
        TrsvGran* gp = gran + (long)bid * NB + r;
        const g2 out = (g2){res, __longlong_as_double(epoch)};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(gp), "v"(out) : "memory");
        if (r < nb) x[r0 + r] = res;
    }
}

// y[r] -= sum_j A[r][j] x[j], j < 64 columns: one wave per row (forward-substitution update)
__global__ __launch_bounds__(256) void gemv_rows_kernel(const double* __restrict__ A, long lda, int rows, int cols,
                                                        const double* __restrict__ x, double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    double s = (lane < cols) ? A[(long)r * lda + lane] * x[lane] : 0.0;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) y[r] -= s;
}

// y[c] -= sum_j A[j][c] x[j], j < rows (<= 64): one lane per column (backward-substitution update)
__global__ __launch_bounds__(256) void gemv_cols_kernel(const double* __restrict__ A, long lda, int rows, int cols,
                                                        const double* __restrict__ x, double* __restrict__ y) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = 0;
    for (; j + 4 <= rows; j += 4) {
        s0 += A[(long)j * lda + c] * x[j];
        s1 += A[(long)(j + 1) * lda + c] * x[j + 1];
        s2 += A[(long)(j + 2) * lda + c] * x[j + 2];
        s3 += A[(long)(j + 3) * lda + c] * x[j + 3];
    }
    for (; j < rows; ++j) s0 += A[(long)j * lda + c] * x[j];
    y[c] -= (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(1024) void dot_kernel(const double* __restrict__ x, const double* __restrict__ y, int n, double* out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += x[i] * y[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        double v = (threadIdx.x < 16) ? red[threadIdx.x] : 0.0;
        for (int o = 8; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (threadIdx.x == 0) *out = v;
    }
}


inline int split(int n, int base = NB) {
    // first part: about half, a multiple of 128 when there is room (keeps GEMM operands aligned and tiles full);
    // with 256-row strip solves at the bottom of the recursion, a multiple of the strip height
    if (base == SB) {
        const int n1 = ((n / 2 + SB - 1) / SB) * SB;
        return n1 < n ? n1 : SB;
    }
    const int q = (n > 256) ? 128 : NB;
    int n1 = ((n / 2 + q - 1) / q) * q;
    if (n1 >= n) n1 = ((n / 2 + NB - 1) / NB) * NB;
    if (n1 >= n) n1 = NB;
    return n1;
}

}  // namespace

int gpk_i_trsm_left(gpk_handle h, bool trans, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
    if (n <= 0 || nrhs <= 0) return 0;
    if (!trans && h->tune.strip && n <= SB && (n & 15) == 0) {
        trsm_strip_kernel<<<gpk_ceil_div(nrhs, SNC), SNT, 0, h->stream>>>(L, ldl, n, B, ldb, nrhs, h->tune.dbg);
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    if (n <= NB) {
        dim3 grid(gpk_ceil_div(nrhs, NB));
        if (trans) trsm_base_kernel<true, false><<<grid, 256, 0, h->stream>>>(L, ldl, n, B, ldb, nrhs, h->tune.dbg);
        else       trsm_base_kernel<false, false><<<grid, 256, 0, h->stream>>>(L, ldl, n, B, ldb, nrhs, h->tune.dbg);
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    const int n1 = split(n, (!trans && h->tune.strip && n > SB) ? SB : NB), n2 = n - n1;
    const double* L21 = L + (long)n1 * ldl;
    const double* L22 = L21 + n1;
    double* B2 = B + (long)n1 * ldb;
    if (!trans) {
        GPK_TRY(gpk_i_trsm_left(h, false, L, n1, ldl, B, nrhs, ldb));
        GPK_TRY(gpk_i_gemm(h, false, false, n2, nrhs, n1, -1.0, L21, ldl, B, ldb, 1.0, B2, ldb, false));
        GPK_TRY(gpk_i_trsm_left(h, false, L22, n2, ldl, B2, nrhs, ldb));
    } else {
        GPK_TRY(gpk_i_trsm_left(h, true, L22, n2, ldl, B2, nrhs, ldb));
        GPK_TRY(gpk_i_gemm(h, true, false, n1, nrhs, n2, -1.0, L21, ldl, B2, ldb, 1.0, B, ldb, false));
        GPK_TRY(gpk_i_trsm_left(h, true, L, n1, ldl, B, nrhs, ldb));
    }
    return 0;
}

// Forward solve L X = B when column c of B (c < lead) is known to be zero in the rows above (lead - 1 - c): the
// Gauss-Newton right-hand side [A(z) | F] with the unknowns stored in REVERSE order has exactly this shape (column j of
// A is zero above row j).  For the sub-problem on rows [row0, row0 + n) only the columns
// c >= lead - (row0 + n) can be non-zero, so every recursive solve, update GEMM and diagonal solve is restricted to
// that contiguous range (rounded down to 64 columns to keep tiles and vector loads aligned).  Columns >= lead are dense.
int gpk_i_trsm_left_lz(gpk_handle h, const double* L, int n, int ldl, double* B, int nrhs, int ldb, int lead, int row0) {
    if (n <= 0 || nrhs <= 0) return 0;
    const int sd = h->lead_div > 0 ? h->lead_div : 1;                // staircase slope 1/sd: column c is zero above row (lead-1-c)/sd
    int clo = lead - sd * (row0 + n);
    clo = clo > 0 ? (clo / NB) * NB : 0;
    if (clo >= nrhs) return 0;
    if (h->tune.strip && n <= SB && (n & 15) == 0) {
        trsm_strip_kernel<<<gpk_ceil_div(nrhs - clo, SNC), SNT, 0, h->stream>>>(L, ldl, n, B + clo, ldb, nrhs - clo, h->tune.dbg);
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    if (n <= NB) {
        trsm_base_kernel<false, false><<<gpk_ceil_div(nrhs - clo, NB), 256, 0, h->stream>>>(L, ldl, n, B + clo, ldb, nrhs - clo, h->tune.dbg);
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    const int n1 = split(n, (h->tune.strip && n > SB) ? SB : NB), n2 = n - n1;
    const double* L21 = L + (long)n1 * ldl;
    double* B2 = B + (long)n1 * ldb;
    GPK_TRY(gpk_i_trsm_left_lz(h, L, n1, ldl, B, nrhs, ldb, lead, row0));
    int c1 = lead - sd * (row0 + n1);                                // X[rows of part 1] is zero left of this column
    c1 = c1 > 0 ? (c1 / NB) * NB : 0;
    if (c1 < nrhs) {                                                 // X1[k][c] is zero for row0 + k < (lead-1-c)/sd: late K start per tile
        const int lz = lead - sd * row0 - c1;
        GPK_TRY(gpk_i_gemm(h, false, false, n2, nrhs - c1, n1, -1.0, L21, ldl, B + c1, ldb, 1.0, B2 + c1, ldb, false, lz > 0 ? lz : 0));
    }
    GPK_TRY(gpk_i_trsm_left_lz(h, L21 + n1, n2, ldl, B2, nrhs, ldb, lead, row0 + n1));
    return 0;
}

// ---- multi-right-hand-side forward solve through explicit inverses of the diagonal blocks ---------------------------------
// The factor L of Theta is fixed for the whole Gauss-Newton iteration while S = L^{-1}[A | F] is recomputed every step, and
// in that solve the only work that is not a GEMM is the 256-row strip substitution: a latency chain (33 strips x 33 us =
// 1.1 ms of the 4.3 ms TRSM phase at BASELINE config 2).  gpk_trtri_diag computes, ONCE per factor and by the same true
// substitution (the substitution solve on an identity right-hand side), the inverses of the db x db diagonal blocks (db = 256,
// 512 or 1024); the solve of a block row then is X_k = inv(L_kk) B_k, a GEMM with a lower-triangular operand (K loop cut at the
// diagonal per row tile), and the whole TRSM runs on the matrix cores.  Larger blocks replace the small, launch-bound GEMMs at
// the bottom of the recursion by one well-filled triangular product (same flops): 3.64 -> see DESIGN.md for db = 1024.  This is what MAGMA / rocBLAS do for trsm; it is only conditionally stable, so it was
// checked on this problem class before being adopted: the residual of inv(L_kk) B_k is bounded by eps cond(L_kk) |B_k| instead
// of eps |L_kk| |X_k|, which is the same size whenever X is large relative to B -- the case here (|L^{-1}A| ~ 1e8).  Measured
// with the CPU oracle at nugget 1e-13 (cond(L) = 1.8e9, diagonal blocks up to 1.5e7): Gauss-Newton iterates differ from the
// substitution path by 8e-14 / 3e-13 / 3e-10 (config 1: N = 1924, db = 256 / 512 / 1024 -- the last is more than half of the
// matrix, block condition 1.7e9) and 5e-14 / 1.3e-13 / 2.6e-13 (config 2: N = 8400, blocks up to 2.9e7), against a parity bound
// of 1e-6 and 5e-9 between equally valid fp64 implementations.  Only the diagonal BLOCKS are inverted; L never is.
// Out of place: the leaf writes X_k = inv(L_kk) B_k into a second buffer (a GEMM cannot run in place over its K range), the
// updates B_2 -= L_21 X_1 read X and modify B, so on return X holds the solution and B is scratch.
__global__ void dinv_identity_kernel(double* __restrict__ D, int n, int db) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;             // one thread per entry of the n x db array
    if (i < (long)n * db) D[i] = ((i / db) % db == i % db) ? 1.0 : 0.0;
}

inline bool dinv_block_ok(int db) { return db == 256 || db == 512 || db == 1024 || db == 2048; }

int gpk_i_trtri_diag(gpk_handle h, const double* L, int n, int ldl, double* Dinv, int db) {
    if (n <= 0) return 0;
    if (!dinv_block_ok(db)) return gpk_bad_arg(h, "trtri_diag: block size must be 256, 512, 1024 or 2048");
    dinv_identity_kernel<<<(unsigned)(((long)n * db + 255) / 256), 256, 0, h->stream>>>(Dinv, n, db);
    GPK_LAUNCH_CHECK(h);
    for (int k0 = 0; k0 < n; k0 += db) {
        const int nk = (n - k0 < db) ? n - k0 : db;
        // (substitution keeps the zeros above the diagonal of the inverse exact: x_i = (0 - sum 0) / l_ii)
        GPK_TRY(gpk_i_trsm_left(h, false, L + (long)k0 * ldl + k0, nk, ldl, Dinv + (long)k0 * db, nk, db));
    }
    return 0;
}

int gpk_i_trsm_left_dinv(gpk_handle h, const double* L, const double* Dinv, int db, int n, int ldl, double* B, int ldb,
                         double* X, int ldx, int nrhs, int lead, int row0) {
    if (n <= 0 || nrhs <= 0) return 0;
    const int sd = h->lead_div > 0 ? h->lead_div : 1;                // staircase slope 1/sd: column c is zero above row (lead-1-c)/sd
    // piecewise profile (gpk_ctx::stair, Darcy): columns and rows are those of this call's top level (column 0 of B, row 0 of L); every
    // product below is told where its column 0 / its k = 0 lie in that frame, and `lead` only says "there is a profile"
    const bool pw = lead > 0 && h->stair.nseg > 0;
    const int base = pw ? h->stair_base : 0;                         // (round 6: B's column 0 is column `base` of the profile's frame -- a column shard)
    int clo = pw ? gpk_stair_first_col(h->stair, row0 + n, base + nrhs) - base : lead - sd * (row0 + n);   // (lead = 0: dense right-hand sides)
    clo = clo > 0 ? (clo / NB) * NB : 0;
    if (clo >= nrhs) return 0;
    if (n <= db) {
        const int lz = pw ? 1 : lead - sd * row0 - clo;
        h->stair_col0 = base + clo; h->stair_row0 = row0;
        return gpk_i_gemm(h, false, false, n, nrhs - clo, n, 1.0, Dinv + (long)row0 * db, db, B + clo, ldb, 0.0, X + clo, ldx,
                          false, lz > 0 ? lz : 0, true);
    }
    int n1 = ((n / 2 + db - 1) / db) * db;                           // first part: about half, a multiple of the block size
    if (n1 >= n) n1 = db;
    const int n2 = n - n1;
    const double* L21 = L + (long)n1 * ldl;
    GPK_TRY(gpk_i_trsm_left_dinv(h, L, Dinv, db, n1, ldl, B, ldb, X, ldx, nrhs, lead, row0));
    int c1 = pw ? gpk_stair_first_col(h->stair, row0 + n1, base + nrhs) - base : lead - sd * (row0 + n1);   // X[rows of part 1] is zero left of this column
    c1 = c1 > 0 ? (c1 / NB) * NB : 0;
    if (c1 < nrhs) {
        const int lz = pw ? 1 : lead - sd * row0 - c1;
        h->stair_col0 = base + c1; h->stair_row0 = row0;
        if (h->tune.solve_splitk && gpk_i_splitk_reserve(h) == 0) {
            // launches that fill the chip badly (a fraction of a wave, or 1.2 waves): more, shorter workgroups (split-K)
            const long t64 = (long)gpk_ceil_div(n2, 64) * gpk_ceil_div(nrhs - c1, 64);
            const bool small = t64 < 2 * h->num_cu;                  // (gpk_i_gemm then uses 32-row tiles, 5 per CU)
            const long tiles = small ? (long)gpk_ceil_div(n2, 32) * gpk_ceil_div(nrhs - c1, 64) : t64;
            const long slots = (long)h->num_cu * (small ? 5 : 4);
            int best = 1; double bc = 1e30;
            for (int s2 = 1; s2 <= h->tune.solve_splitk; ++s2) {
                const double c = (double)((tiles * s2 + slots - 1) / slots) / s2 + 0.04 * (s2 - 1);
                if (c < bc - 1e-9) { bc = c; best = s2; }
            }
            h->splitk_req = best;
        }
        const int rc = gpk_i_gemm(h, false, false, n2, nrhs - c1, n1, -1.0, L21, ldl, X + c1, ldx, 1.0, B + (long)n1 * ldb + c1, ldb,
                                  false, lz > 0 ? lz : 0);
        h->splitk_req = 0;
        GPK_TRY(rc);
    }
    return gpk_i_trsm_left_dinv(h, L21 + n1, Dinv, db, n2, ldl, B + (long)n1 * ldb, ldb, X + (long)n1 * ldx, ldx, nrhs, lead, row0 + n1);
}

// The 64-row diagonal solves of a multi-RHS TRSM keep only nrhs/64 waves busy and sit on the critical path between
// the GEMM updates.  Column groups of the right-hand side are independent, so they are issued on separate streams:
// while one group runs a (latency-bound) diagonal solve the other groups' GEMMs fill the chip.
int gpk_i_trsm_left_mt(gpk_handle h, bool trans, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
    constexpr int G = 4;
    // Measured at N=8400, nrhs=4001 (round 1): 17.9 ms with 4 groups vs 11.0 ms single-stream -- four times as many
    // launches of smaller GEMMs cost more than the overlap buys.  Kept behind gpk_debug_set(2, 1) for re-evaluation
    // once the diagonal solves are fused into fewer launches.
    if (!h->tune.mt_trsm || nrhs < 1024 || n <= 2 * NB) return gpk_i_trsm_left(h, trans, L, n, ldl, B, nrhs, ldb);
    if (!h->ev_fork) {
        GPK_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < G - 1; ++i) {
            GPK_HIP(h, hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking));
            GPK_HIP(h, hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
    }
    const int per = ((nrhs / G + 127) / 128) * 128;                  // group width, multiple of 128 columns
    hipStream_t main_stream = h->stream;
    GPK_HIP(h, hipEventRecord(h->ev_fork, main_stream));
    int rc = 0;
    for (int g = 0; g < G && rc == 0; ++g) {
        const int c0 = g * per;
        if (c0 >= nrhs) break;
        const int w = (nrhs - c0 < per || g == G - 1) ? nrhs - c0 : per;
        if (g > 0) {
            h->stream = h->side[g - 1];
            hipError_t e = hipStreamWaitEvent(h->stream, h->ev_fork, 0);
            if (e != hipSuccess) { h->stream = main_stream; return gpk_fail(h, e, "hipStreamWaitEvent", __FILE__, __LINE__); }
        }
        rc = gpk_i_trsm_left(h, trans, L, n, ldl, B + c0, w, ldb);
        if (g > 0) {
            hipError_t e = hipEventRecord(h->ev_join[g - 1], h->stream);
            h->stream = main_stream;
            if (e == hipSuccess) e = hipStreamWaitEvent(main_stream, h->ev_join[g - 1], 0);
            if (e != hipSuccess) return gpk_fail(h, e, "join", __FILE__, __LINE__);
        }
    }
    h->stream = main_stream;
    return rc;
}

int gpk_i_trsm_right_lt(gpk_handle h, const double* L, int n, int ldl, double* X, int m, int ldx) {
    if (n <= 0 || m <= 0) return 0;
    if (n <= NB) {
        trsm_base_kernel<false, true><<<gpk_ceil_div(m, NB), 256, 0, h->stream>>>(L, ldl, n, X, ldx, m, h->tune.dbg);
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    const int n1 = split(n), n2 = n - n1;
    const double* L21 = L + (long)n1 * ldl;
    const double* L22 = L21 + n1;
    double* X2 = X + n1;
    GPK_TRY(gpk_i_trsm_right_lt(h, L, n1, ldl, X, m, ldx));
    GPK_TRY(gpk_i_gemm(h, false, true, m, n2, n1, -1.0, X, ldx, L21, ldl, 1.0, X2, ldx, false));
    GPK_TRY(gpk_i_trsm_right_lt(h, L22, n2, ldl, X2, m, ldx));
    return 0;
}

// Two-level right-looking Cholesky.
//   outer panels of OB columns: after a panel is factored, ONE large SYRK (K = OB, lower tiles only) updates the
//   trailing matrix on the MFMA units;
//   inside a panel, steps of NB = 64 columns, three launches each, every one covering ALL rows of the panel:
//     potf2 (diagonal block)  ->  substitution of the rows below  ->  rank-64 update of the panel's remaining columns.
// (A plain recursion needs ~15 launches per 64 columns, most of them a single wave.)
// Factor a tall block column: D = A[0:ob, 0:ob] is replaced by its Cholesky factor and the rows below by A[ob:, 0:ob] D^{-T}
// (64-column panel kernel + rank-64 update of the remaining columns, for all nrows rows).  The building block of both
// the single-GPU factorisation below and the panel-sharded multi-GPU one (gpk/sharded.py, the owner's share of a step).
int gpk_i_potrf_panel(gpk_handle h, double* A, int nrows, int ob, int lda, int pivot_base, bool left_looking, void* ev_wait_p1, void* ev_rec_pre) {
    // ev_wait_p1 (pipeline): the columns from 64 on become valid only with this event -- waited for before the second panel;
    // ev_rec_pre: recorded once every panel but the last has been factored (the pipeline starts the next block's update with those)
    if (ob <= 0 || nrows < ob) return 0;
    const int npan = gpk_ceil_div(ob, NB);
    // FUSED schedule (h->tune.panel_fused, default): the rank-64 work between two panels rides inside the panel kernels (PanelFuse above)
    // Right-looking schedule (whole chip) only: on the 32-CU chain partition of the pipelined phase the product workgroups sit on the
    // same CUs as the panel workgroups and slow them down (co-residence, the very reason for the CU partition): measured 3.13 -> 3.26 ms
    // for the phase at config 2 with the left-looking chain fused, so that chain keeps its separate launches (h->tune.panel_fused = 2 forces it).
    const bool fused = h->tune.panel_fused && (h->tune.panel_fused == 2 || !left_looking) && h->tune.fused_panel && h->tune.panel_mfma == 1 && h->tune.panel_unrolled && !ev_wait_p1 && !ev_rec_pre;
    for (int j0 = 0; j0 < ob; j0 += NB) {
        const int nb = (ob - j0 < NB) ? ob - j0 : NB;
        double* Ajj = A + (long)j0 * lda + j0;
        const int below = nrows - (j0 + nb);
        if (fused) {
            const int nrb = below > 0 ? gpk_ceil_div(below, NB) : 0;
            const unsigned target = h->panel_loaded + (unsigned)nrb;
            PanelFuse f;
            memset(&f, 0, sizeof f);
            f.npanel = 1 + nrb;
            f.prevK = j0 > 0 ? NB : 0;                              // part B: the previous panel applied to this panel's columns
            const int r1 = j0 + NB;                                  // first row / column behind this panel
            if (j0 > 0 && r1 < ob && r1 < nrows) {
                // part A (independent of this panel).  Left-looking: the block's panels 0 .. j-1 (all final) applied to the NEXT panel's
                // columns, which then only lack panel j -- part B of the next kernel;
                // right-looking: panel j-1 applied to all remaining columns of the block.  Rows from r1 down (rows above belong to the
                // upper triangle of the block).
                f.gm = nrows - r1;
                f.gC = A + (long)r1 * lda + r1;
                if (left_looking) { f.gn = std::min(NB, ob - r1); f.gK = j0;      f.gA = A + (long)r1 * lda; }
                else              { f.gn = ob - r1;              f.gK = NB;      f.gA = A + (long)r1 * lda + (j0 - NB); }
                f.gB = f.gA;                                         // rows r1 .. r1 + gn - 1 of the same columns
                f.gtn = gpk_ceil_div(f.gn, NB);
                if (f.gK <= 0 || f.gn <= 0 || f.gm <= 0) { f.gm = f.gn = f.gK = f.gtn = 0; }
            }
            const int extra = f.gK > 0 ? gpk_ceil_div(f.gm, PFM) * f.gtn : 0;
            potrf_panel_mfma_kernel<8, true, true><<<1 + nrb + extra, 256, 0, h->stream>>>(Ajj, lda, nb, below, h->d_info, pivot_base + j0,
                                                                                            (unsigned*)(h->d_flags + GPK_MAX_TRSV_BLOCKS), target, h->tune.dbg, f);
            GPK_LAUNCH_CHECK(h);
            h->panel_loaded = target;
            continue;
        }
        // (left-looking: the other columns are first touched by the second panel's update; right-looking: by the first panel's)
        if (ev_wait_p1 && j0 == (left_looking ? NB : 0)) GPK_HIP(h, hipStreamWaitEvent(h->stream, (hipEvent_t)ev_wait_p1, 0));
        if (left_looking && j0 > 0) {
            // LEFT-looking inside the block column (used when the chain runs on the small CU partition of the pipeline): the
            // 64 columns of this panel receive the contributions of all earlier panels of the block in one product with K = j0
            // <= 448 and rows/32 workgroups, instead of every panel pushing a rank-64 update into all remaining columns (up to
            // 7 x rows/32 workgroups, which a 32-CU partition works off in several rounds: 20-30 us instead of 9).  Same flops.
            // On the whole chip the right-looking form is faster (the long-K product on ~110 workgroups is latency-bound:
            // 4.42 vs 4.25 ms for the phase), so it remains the default everywhere else.
            // (Tried in round 2: the contributions of the panels up to p-2 on a SECOND stream of the same partition, next to panel
            // p-1's kernel, so that only a rank-64 update stays between two panel kernels.  Not possible on this runtime: a fourth
            // stream in use shares a hardware queue with one of the others -- CU mask included -- see pipe_setup.  Round 3: the same
            // idea INSIDE the panel kernel, see PanelFuse.)
            GPK_TRY(gpk_i_gemm(h, false, true, nrows - j0, nb, j0, -1.0, A + (long)j0 * lda, lda, A + (long)j0 * lda, lda, 1.0, Ajj, lda, false));
        }
        {
            const int nrb = below > 0 ? gpk_ceil_div(below, NB) : 0;
            const unsigned target = h->panel_loaded + (unsigned)nrb;
            PanelFuse nofuse;
            memset(&nofuse, 0, sizeof nofuse);
            if (h->tune.panel_unrolled)
                potrf_panel_mfma_kernel<8, true><<<1 + nrb, 256, 0, h->stream>>>(Ajj, lda, nb, below, h->d_info, pivot_base + j0,
                                                                                  (unsigned*)(h->d_flags + GPK_MAX_TRSV_BLOCKS), target, h->tune.dbg, nofuse);
            else
                potrf_panel_mfma_kernel<8, false><<<1 + nrb, 256, 0, h->stream>>>(Ajj, lda, nb, below, h->d_info, pivot_base + j0,
                                                                                   (unsigned*)(h->d_flags + GPK_MAX_TRSV_BLOCKS), target, h->tune.dbg, nofuse);
            GPK_LAUNCH_CHECK(h);                                     // a failed launch issues no tickets: count them only now
            h->panel_loaded = target;
        }
        if (ev_rec_pre && j0 / NB == npan - 2) GPK_HIP(h, hipEventRecord((hipEvent_t)ev_rec_pre, h->stream));   // (this panel's columns are final)
        if (below > 0 && !left_looking) {
            double* Abj = A + (long)(j0 + nb) * lda + j0;
            const int pc = ob - (j0 + nb);                           // remaining columns of this block column
            if (pc > 0) {
                // A[j0+nb:, j0+nb : ob] -= L[j0+nb:, j] * L[j0+nb : ob, j]^T   (rows above the diagonal of that block are
                // computed too; they are never read)
                GPK_TRY(gpk_i_gemm(h, false, true, below, pc, nb, -1.0, Abj, lda, Abj, lda, 1.0,
                                   A + (long)(j0 + nb) * lda + (j0 + nb), lda, false));
            }
        }
    }
    GPK_LAUNCH_CHECK(h);
    return 0;
}

static int potrf_pipelined(gpk_handle h, const double* W, int ldw, int rows, int nc, int lead, double* Hb, int ldh, double* d_loss,
                           int pivot_base);

static int potrf_seq(gpk_handle h, double* A, int n, int lda, int pivot_base) {
    if (n <= 0) return 0;
    const int OB = (h->tune.potrf_ob >= 64 && h->tune.potrf_ob % 64 == 0) ? h->tune.potrf_ob : 512;
    for (int k0 = 0; k0 < n; k0 += OB) {
        const int ob = (n - k0 < OB) ? n - k0 : OB;
        // (gpk_tune key 56, round-6 experiment: the left-looking form inside the block on the whole chip -- with the fused schedule forced,
        // key 48 = 2, the extra workgroups of panel kernel j apply ALL finished panels of the block to the next panel's 64 columns (K = 64 j,
        // rows/64 workgroups) instead of panel j-1 to all remaining columns (K = 64, up to 7 rows/64 workgroups))
        GPK_TRY(gpk_i_potrf_panel(h, A + (long)k0 * lda + k0, n - k0, ob, lda, pivot_base + k0, h->tune.seq_left_looking != 0));
        const int rest = n - (k0 + ob);
        if (rest > 0) {
            double* P = A + (long)(k0 + ob) * lda + k0;               // factored panel rows below the outer block
            GPK_TRY(gpk_i_gemm(h, false, true, rest, rest, ob, -1.0, P, lda, P, lda, 1.0,
                               A + (long)(k0 + ob) * lda + (k0 + ob), lda, true));
        }
    }
    return 0;
}

static int pipe_setup(gpk_handle h, size_t nev, size_t ntev, bool reserve);
#ifdef GPK_DEV                                                       // measured and not adopted (round 5): development build only
// RIGHT-looking factorisation with look-ahead on the two CU partitions (round 5; gpk_tune key 54 = 1 selects it for the orders of keys 19 /
// 20 instead of the left-looking pipeline above).  MEASURED, NOT ADOPTED (profiles/r05_potrf_lookahead.txt, tools/potrf_modes_probe.py): bit-identical
// factor, but at order 8400 7.4 ms with a 64-CU chain partition / 9.1 ms with 32 CUs against 6.6 ms on one stream -- the tall fused panel kernels
// run 1.75x / 2.5x longer on the partition (41.7 / 60.4 us instead of 23.8: two to three rounds of workgroups) and become the critical path while
// the products lose a quarter / an eighth of the chip; 9600: 8.83 vs 8.78; 16000: 30.1 vs 29.7.  potrf_seq leaves the chip nearly idle during the panel chains (8 fused panel kernels of
// ~22 us per 512 columns: 40 % of the factorisation at order 9600) and the left-looking pipeline pays for its overlap with skinny, long-K
// block-column updates.  Here the trailing update stays the large rank-512 product it is in potrf_seq, cut in two: on the GEMM partition,
// block k's update of the NEXT block column first (rows x 512 x 512), then -- while the chain partition already factors that column -- its
// update of everything to the right.  Per block: chain_k (C) -> next-column update (G) -> chain_{k+1} (C) || rest of update k (G).
static int potrf_lookahead(gpk_handle h, double* A, int n, int lda, int pivot_base) {
    const int OB = (h->tune.potrf_ob >= 64 && h->tune.potrf_ob % 64 == 0) ? h->tune.potrf_ob : 512;
    const int J = gpk_ceil_div(n, OB);
    if (J < 3) return potrf_seq(h, A, n, lda, pivot_base);
    GPK_TRY(pipe_setup(h, 2 * (size_t)J + 1, 0, true));
    const hipStream_t main_s = h->stream, G = h->pipe_g, C = h->pipe_c;
    hipEvent_t* ev_col = h->pipe_ev.data();                          // [J]: block column j carries every update of the blocks before it
    hipEvent_t* ev_chain = h->pipe_ev.data() + J;                    // [J]: block column j is factored
    hipEvent_t ev_fork = h->pipe_ev[2 * J];
    auto fail = [&](hipError_t e, const char* what) {
        h->stream = main_s; h->no_sk = 0;
        (void)hipStreamSynchronize(G); (void)hipStreamSynchronize(C);
        return gpk_fail(h, e, what, __FILE__, __LINE__);
    };
#define LA_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(e__, #call); } while (0)
    // block column 0: nothing to overlap with -- on the whole chip, before the fork
    int rc = gpk_i_potrf_panel(h, A, n, std::min(OB, n), lda, pivot_base);
    if (rc) return rc;
    LA_HIP(hipEventRecord(ev_fork, main_s));
    LA_HIP(hipStreamWaitEvent(G, ev_fork, 0));
    LA_HIP(hipStreamWaitEvent(C, ev_fork, 0));
    for (int j = 0; j < J && rc == 0; ++j) {
        const int k0 = j * OB, ob = std::min(OB, n - k0), k1 = k0 + ob, rest = n - k1;
        if (j > 0) {                                                 // chain of block column j on the chain partition
            h->stream = C; h->no_sk = 1;
            LA_HIP(hipStreamWaitEvent(C, ev_col[j], 0));
            rc = gpk_i_potrf_panel(h, A + (long)k0 * lda + k0, n - k0, ob, lda, pivot_base + k0);
            h->no_sk = 0;
            if (rc) break;
            LA_HIP(hipEventRecord(ev_chain[j], C));
        }
        if (rest <= 0) break;
        h->stream = G;
        if (j > 0) LA_HIP(hipStreamWaitEvent(G, ev_chain[j], 0));
        const double* P = A + (long)k1 * lda + k0;                   // rows k1.. of the factored block column j
        const int ob1 = std::min(OB, rest);
        // the next block column first (its top square is a diagonal block: tiles above the diagonal are not computed) ...
        rc = gpk_i_gemm(h, false, true, rest, ob1, ob, -1.0, P, lda, P, lda, 1.0, A + (long)k1 * lda + k1, lda, false, 0, false, true);
        if (rc) break;
        LA_HIP(hipEventRecord(ev_col[j + 1], G));
        // ... then everything to its right, next to the chain of block column j + 1
        const int rest2 = rest - ob1;
        if (rest2 > 0) {
            const double* P2 = A + (long)(k1 + ob1) * lda + k0;
            rc = gpk_i_gemm(h, false, true, rest2, rest2, ob, -1.0, P2, lda, P2, lda, 1.0, A + (long)(k1 + ob1) * lda + (k1 + ob1), lda, true);
            if (rc) break;
        }
    }
    h->stream = main_s; h->no_sk = 0;
    if (rc) { (void)hipStreamSynchronize(G); (void)hipStreamSynchronize(C); return rc; }
    LA_HIP(hipEventRecord(ev_fork, G));                              // (re-used: everything the GEMM partition was given)
    LA_HIP(hipStreamWaitEvent(main_s, ev_fork, 0));
    LA_HIP(hipStreamWaitEvent(main_s, ev_chain[J - 1], 0));
#undef LA_HIP
    return 0;
}

#endif

int gpk_i_potrf(gpk_handle h, double* A, int n, int lda, int pivot_base) {
    if (h->tune.pipeline && !h->pipe_unavailable && n >= h->tune.potrf_pipeline_min_n && n <= h->tune.potrf_pipeline_max_n && h->num_cu >= 64)
    {
#ifdef GPK_DEV
        if (h->tune.potrf_lookahead) return potrf_lookahead(h, A, n, lda, pivot_base);
#endif
        return potrf_pipelined(h, nullptr, 0, 0, n, 0, A, lda, nullptr, pivot_base);
    }
    return potrf_seq(h, A, n, lda, pivot_base);
}

// ---- SYRK + Cholesky of the Gauss-Newton matrix, pipelined on two CU partitions -------------------------------------------
// Hb = W^T W followed by chol(Hb) is 1.6 ms of GEMM followed by a 2.5 ms phase that is mostly a latency chain (63 panel
// kernels of ~25 us that keep <= 63 workgroups busy, plus their rank-64 updates).  Running the two on plain concurrent
// streams does not overlap them: the GEMM's workgroups fill every CU, and a panel workgroup that does squeeze in next to
// GEMM waves runs 3-8x slower (measured, tools/overlap_probe.py: issue arbitration, LDS and the matrix pipe are shared; a
// raised s_setprio does not help).  What works is a SPATIAL partition: hipExtStreamCreateWithCUMask gives the chain stream c
// CUs (bit i of the mask is a CU of XCD i mod 8 -- tools/cu_mask_probe.py -- so a multiple of 8 takes c/8 CUs from every
// XCD) and the GEMM stream the other 256 - c, and the factorisation is reordered LEFT-looking over 512-column blocks so that
// it can start before the product is complete:
//     GEMM stream, block j:  S_j: Hb[j0:, j0:j0+512] = W[:, j0:]^T W[:, j0:j0+512]          (all rows below, leading zeros skipped)
//                            U_j: Hb[j0:, j0:j0+512] -= L[j0:, 0:j0] L[j0:j0+512, 0:j0]^T   (panels 0..j-2 first, then -- once the
//                                 chain stream has finished block j-1 -- the last 512 columns)
//     chain stream, block j: wait for U_j; gpk_i_potrf_panel (64-column panel kernels + rank-64 updates inside the block)
// The chain of block j runs next to S_{j+1} and the early part of U_{j+1}.  Same flops as the right-looking order (plus the
// upper triangles of the 512 x 512 diagonal blocks, which are computed and never read).
// The partition costs the GEMM side a quarter of the chip, the chain side roughly doubles the time of its rank-64 updates;
// the overlap pays while the panel chain (~34 us per 64 columns) is comparable to the GEMM work (~n^3): measured 4.24 -> 3.96 ms
// at n = 4001 (BASELINE config 2) but 35.9 -> 43.8 ms at n = 10001, break-even near n = 5500.

// Block columns of the pipelined factorisation: a first block of h->tune.pipeline_w0 columns (the chain can only start once its product is
// there), then blocks of h->tune.pipeline_ob columns; widths are multiples of the panel width, at most 512.
static std::vector<int> pipe_blocks(gpk_handle h, int nc) {
    auto norm = [](int w) { w = (w / NB) * NB; return w < NB ? NB : (w > 512 ? 512 : w); };
    std::vector<int> b{0};
    int w = norm(h->tune.pipeline_w0);
    while (b.back() < nc) { b.push_back(b.back() + w < nc ? b.back() + w : nc); w = norm(h->tune.pipeline_ob); }
    return b;
}

// The two masked streams are created by gpk_create, right after the handle's own stream, and again here only when the partition
// size is changed (development switch).  The order matters on this runtime (measured, round 2, tools/alias_probe.py): with ONE other
// stream created between the handle's stream and the masked ones -- used or not -- every phase of the step ran slower (solve +14 %,
// pipelined phase 3.6 -> 4.6 ms; two such streams: 10 ms), and so did a FOURTH stream in use (a second chain stream was tried for
// the panels' early updates).  HIP multiplexes streams onto a few hardware queues, and streams that land on the same queue
// serialise; streams created before the context (torch's pool) or after the masked pair did no harm.  GPU_MAX_HW_QUEUES = 8 changed
// nothing.
int gpk_i_pipe_streams(gpk_handle h) { return pipe_setup(h, 0, 0, false); }

static int pipe_setup(gpk_handle h, size_t nev, size_t ntev, bool reserve = true) {
    // multiples of 32: bits 8k .. 8k+7 of the mask are one CU of shader engine k mod 4 on each of the 8 XCDs, so 32 bits take one CU from
    // every shader engine; other sizes leave the engines uneven (measured: 48 behaves like 32, 80 like 64)
    int c = ((h->tune.pipeline_chain_cus + 16) / 32) * 32;
    if (c < 32) c = 32;
    if (c > h->num_cu - 32) c = h->num_cu - 32;
    if (!h->pipe_g || h->pipe_chain_cus != c) {
        if (h->pipe_g) { GPK_HIP(h, hipStreamSynchronize(h->pipe_g)); GPK_HIP(h, hipStreamDestroy(h->pipe_g)); h->pipe_g = nullptr; }
        if (h->pipe_c) { GPK_HIP(h, hipStreamSynchronize(h->pipe_c)); GPK_HIP(h, hipStreamDestroy(h->pipe_c)); h->pipe_c = nullptr; }
        uint32_t mc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < h->num_cu && i < 256; ++i) ((i < c) ? mc : mg)[i >> 5] |= 1u << (i & 31);
        GPK_HIP(h, hipExtStreamCreateWithCUMask(&h->pipe_c, 8, mc));
        GPK_HIP(h, hipExtStreamCreateWithCUMask(&h->pipe_g, 8, mg));
        h->pipe_chain_cus = c;
    }
    if (reserve) GPK_TRY(gpk_i_splitk_reserve(h));                  // split-K products of the GEMM stream (gpk_gemm.hip); not at gpk_create
    while (h->pipe_ev.size() < nev) {
        hipEvent_t e;
        GPK_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->pipe_ev.push_back(e);
    }
    while (h->pipe_tev.size() < ntev) {
        hipEvent_t e;
        GPK_HIP(h, hipEventCreate(&e));
        h->pipe_tev.push_back(e);
    }
    return 0;
}

int gpk_i_syrk_potrf(gpk_handle h, const double* W, int ldw, int rows, int nc, int lead, double* Hb, int ldh, double* d_loss) {
    const int J = gpk_ceil_div(nc, 512) < 3 ? 0 : (int)pipe_blocks(h, nc).size() - 1;   // (orders below 1025: nothing to overlap)
    h->pipe_tev_used = 0;
    h->prof_pipelined = 0;
    // (a handle on which the CU-masked streams could not be created -- a runtime or a container that does not allow CU masks --
    // silently keeps the one-stream schedule: same results, no overlap)
    if (!h->pipe_unavailable && h->tune.pipeline && J >= 3 && nc <= h->tune.pipeline_max_n && h->num_cu >= 64 && pipe_setup(h, (h->tune.pipeline_lookahead ? 4 : 2) * (size_t)J + 1, h->prof ? 2 * (size_t)J : 0) != 0) {
        h->pipe_unavailable = true;
        (void)hipGetLastError();
    }
    if (h->pipe_unavailable || !h->tune.pipeline || J < 3 || nc > h->tune.pipeline_max_n || h->num_cu < 64) {   // small systems: nothing to overlap; large: see above
        if (h->prof) {
            while (h->pipe_tev.size() < 2) { hipEvent_t e; GPK_HIP(h, hipEventCreate(&e)); h->pipe_tev.push_back(e); }
            GPK_HIP(h, hipEventRecord(h->pipe_tev[0], h->stream));
        }
        const int ph = h->prof_phase;
        h->prof_phase = 2;                                           // (flop accounting: the product, apart from the factorisation's updates)
        h->stair_col0 = 0; h->stair_row0 = 0;                        // (a piecewise profile set by the caller, gpk_ctx::stair, is in W's own frame)
        const int rcp = gpk_i_gemm(h, true, false, nc, nc, rows, 1.0, W, ldw, W, ldw, 0.0, Hb, ldh, true, lead);
        h->prof_phase = ph;
        GPK_TRY(rcp);
        if (h->prof) { GPK_HIP(h, hipEventRecord(h->pipe_tev[1], h->stream)); h->pipe_tev_used = 2; }
        if (d_loss) GPK_HIP(h, hipMemcpyAsync(d_loss, Hb + (long)(nc - 1) * ldh + (nc - 1), sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        return gpk_i_potrf(h, Hb, nc, ldh, 0);
    }
    GPK_TRY(potrf_pipelined(h, W, ldw, rows, nc, lead, Hb, ldh, d_loss, 0));
    h->prof_pipelined = 1;
    return 0;
}

// W != nullptr: Hb <- chol(W^T W), the product pipelined with the factorisation (Gauss-Newton step).  W == nullptr: Hb holds a
// symmetric positive definite matrix (lower triangle) and is factored in place -- the same two-partition schedule without the
// products: left-looking 512-column block updates on the GEMM partition, panel chains on the chain partition.
static int potrf_pipelined(gpk_handle h, const double* W, int ldw, int rows, int nc, int lead, double* Hb, int ldh, double* d_loss,
                           int pivot_base) {
    const std::vector<int> bnd = pipe_blocks(h, nc);
    const int J = (int)bnd.size() - 1;
    if (J < 2) return potrf_seq(h, Hb, nc, ldh, pivot_base);
    GPK_TRY(pipe_setup(h, (h->tune.pipeline_lookahead ? 4 : 2) * (size_t)J + 1, h->prof ? 2 * (size_t)J : 0));
    const hipStream_t main_s = h->stream, G = h->pipe_g, C = h->pipe_c;
    // product of the blocks [jb, je): Hb[b_jb :, b_jb : b_je] = W[:, b_jb :]^T W[:, b_jb : b_je] (tiles above the diagonal skipped)
    auto product = [&](int jb, int je) {
        const int j0 = bnd[jb], j1 = bnd[je < J ? je : J];
        // few tiles, long K: split K so that the launch has about h->tune.pipeline_units workgroups (see GemmArgs::splitk)
        const int th = h->tune.pipeline_tile == 128 ? 128 : h->tune.pipeline_tile == 64 ? 64 : 32;
        const long tiles = (long)gpk_ceil_div(nc - j0, th) * gpk_ceil_div(j1 - j0, 64);
        h->splitk_req = h->tune.pipeline_units > 0 ? (int)((h->tune.pipeline_units + tiles / 2) / tiles) : 0;
        h->tile_req = jb > 0 ? h->tune.pipeline_tile : 0;
        h->stair_col0 = j0; h->stair_row0 = 0;                       // (piecewise profile of the caller: this product's column 0 is W's column j0)
        const int r = gpk_i_gemm(h, true, false, nc - j0, j1 - j0, rows, 1.0, W + j0, ldw, W + j0, ldw, 0.0, Hb + (long)j0 * ldh + j0, ldh, false,
                                 lead > j0 ? lead - j0 : 0, false, true);
        h->splitk_req = 0; h->tile_req = 0;
        return r;
    };
    hipEvent_t* ev_ready = h->pipe_ev.data();                        // [J]
    hipEvent_t* ev_chain = h->pipe_ev.data() + J;                    // [J]
    hipEvent_t ev_fork = h->pipe_ev[2 * J];
    hipEvent_t* ev_ready2 = h->pipe_ev.data() + 2 * J + 1;           // [J] (lookahead only)
    hipEvent_t* ev_pre = h->pipe_ev.data() + 3 * J + 1;              // [J] (lookahead only)
    int rc = 0, ntev = 0;
    auto fail = [&](hipError_t e, const char* what) {               // restore the handle's stream and drain both side streams before reporting
        h->stream = main_s; h->no_sk = 0;
        (void)hipStreamSynchronize(G); (void)hipStreamSynchronize(C);
        return gpk_fail(h, e, what, __FILE__, __LINE__);
    };
#define PIPE_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(e__, #call); } while (0)
    auto timed_product = [&](hipStream_t s, int jb, int je) -> int {  // (HIP events around the launch: bench.py's roofline leg)
        if (h->prof) { hipError_t e = hipEventRecord(h->pipe_tev[ntev], s); if (e != hipSuccess) return fail(e, "hipEventRecord"); }
        const int ph = h->prof_phase;
        h->prof_phase = 2;                                           // (flop accounting: the product, apart from the factorisation's updates)
        const int r = product(jb, je);
        h->prof_phase = ph;
        if (h->prof) {
            hipError_t e = hipEventRecord(h->pipe_tev[ntev + 1], s); if (e != hipSuccess) return fail(e, "hipEventRecord");
            ntev += 2; h->pipe_tev_used = ntev;
        }
        if (r == 0 && je >= J && d_loss) {                           // the bordered matrix' last diagonal entry, before any update
            hipError_t e = hipMemcpyAsync(d_loss, Hb + (long)(nc - 1) * ldh + (nc - 1), sizeof(double), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return fail(e, "hipMemcpyAsync");
        }
        return r;
    };
    // The first `pre` blocks of the product run on the whole chip before the fork (one launch): nothing could overlap with
    // block 0.  Measured at config 2 (phase time, ms): pre = 1 / 2 / 3 -> 4.01 / 4.16 / 4.37 with a 64-CU chain partition; with
    // the left-looking chain, pre = 1 and chain partitions of 16 / 24 / 32 / 40 / 64 CUs -> 4.65 / 4.06 / 3.88 / 3.98 / 4.10;
    // sequential (one stream, whole chip) 4.25.
    int pre = h->tune.pipeline_pre < 1 ? 1 : h->tune.pipeline_pre;
    if (pre > J - 1) pre = J - 1;
    if (W) {
        rc = timed_product(main_s, 0, pre);
        if (rc) return rc;
    }
    PIPE_HIP(hipEventRecord(ev_fork, main_s));
    PIPE_HIP(hipStreamWaitEvent(G, ev_fork, 0));
    PIPE_HIP(hipStreamWaitEvent(C, ev_fork, 0));
    for (int j = 0; j < J && rc == 0; ++j) {
        const int j0 = bnd[j], ob = bnd[j + 1] - j0, m = nc - j0;
        const int pb0 = j > 0 ? bnd[j - 1] : 0, pob = j0 - pb0;      // the previous block: first column, width
        double* Hjj = Hb + (long)j0 * ldh + j0;
        h->stream = G;
        if (j > 0) {
            const double* Lrow = Hb + (long)j0 * ldh;                // rows j0.. of the factored panels, columns 0..j0
            if (j > 1) {                                             // panels 0..j-2: their chain finished an iteration ago
                const long tiles = (long)gpk_ceil_div(m, 32) * gpk_ceil_div(ob, 64);
                h->splitk_req = h->tune.pipeline_units > 0 ? (int)((h->tune.pipeline_units + tiles / 2) / tiles) : 0;
                rc = gpk_i_gemm(h, false, true, m, ob, pb0, -1.0, Lrow, ldh, Lrow, ldh, 1.0, Hjj, ldh, false, 0, false, true);
                h->splitk_req = 0;
                if (rc) break;
            }
            // ... then block j-1, whose chain is still running: its first seven panels as soon as they are final (ev_pre), the last
            // one (rank 64) when the chain has finished -- first into the 64 columns the chain of block j starts with (ev_ready),
            // then into the other columns (ev_ready2, needed from the second panel on).  What lies between two block chains is
            // two event hops and one small rank-64 launch instead of a K = 512 product over the whole block column.
            const double* Lb = Lrow + pb0;
            const int k1 = (h->tune.pipeline_lookahead && pob > NB) ? pob - NB : 0;
            if (k1 > 0) {
                PIPE_HIP(hipStreamWaitEvent(G, ev_pre[j - 1], 0));
                rc = gpk_i_gemm(h, false, true, m, ob, k1, -1.0, Lb, ldh, Lb, ldh, 1.0, Hjj, ldh, false, 0, false, true);
                if (rc) break;
            }
            PIPE_HIP(hipStreamWaitEvent(G, ev_chain[j - 1], 0));
            if (h->tune.pipeline_lookahead) {
                const int w0 = ob < NB ? ob : NB;
                rc = gpk_i_gemm(h, false, true, m, w0, pob - k1, -1.0, Lb + k1, ldh, Lb + k1, ldh, 1.0, Hjj, ldh, false);
                if (rc) break;
                PIPE_HIP(hipEventRecord(ev_ready[j], G));
                if (ob > w0) {
                    rc = gpk_i_gemm(h, false, true, m, ob - w0, pob - k1, -1.0, Lb + k1, ldh, Lb + k1 + (long)w0 * ldh, ldh, 1.0, Hjj + w0, ldh, false);
                    if (rc) break;
                }
                PIPE_HIP(hipEventRecord(ev_ready2[j], G));
            } else {
                rc = gpk_i_gemm(h, false, true, m, ob, pob, -1.0, Lb, ldh, Lb, ldh, 1.0, Hjj, ldh, false, 0, false, true);
                if (rc) break;
            }
        }
        if (j == 0 || !h->tune.pipeline_lookahead) PIPE_HIP(hipEventRecord(ev_ready[j], G));
        if (W && j + pre < J) {                                      // product of the block `pre` ahead, while the chain of block j runs
            rc = timed_product(G, j + pre, j + pre + 1);
            if (rc) break;
        }
        h->stream = C;
        h->no_sk = 1;                                                // (the tile-list workspace belongs to the GEMM stream while both run)
        PIPE_HIP(hipStreamWaitEvent(C, ev_ready[j], 0));
        rc = gpk_i_potrf_panel(h, Hjj, m, ob, ldh, pivot_base + j0, h->tune.left_looking_panels != 0,
                               (h->tune.pipeline_lookahead && j > 0) ? ev_ready2[j] : nullptr, (h->tune.pipeline_lookahead && j + 1 < J && ob > NB) ? ev_pre[j] : nullptr);
        h->no_sk = 0;
        if (rc) break;
        PIPE_HIP(hipEventRecord(ev_chain[j], C));
    }
    h->stream = main_s;
    h->no_sk = 0;
    if (rc) {                                                        // drain both side streams before reporting
        (void)hipStreamSynchronize(G); (void)hipStreamSynchronize(C);
        return rc;
    }
    PIPE_HIP(hipStreamWaitEvent(main_s, ev_chain[J - 1], 0));       // (everything on G precedes ev_ready[J-1], which C waited for)
#undef PIPE_HIP
    return 0;
}


int gpk_i_trsv(gpk_handle h, bool trans, const double* L, int n, int ldl, double* x) {
    if (n <= 0) return 0;
    const int nblk = gpk_ceil_div(n, NB);
    if (h->tune.fused_trsv == 1 && nblk <= GPK_MAX_TRSV_BLOCKS) {     // data-tagged hand-offs (default)
        // (trsv_alt: a solve issued on a secondary stream NEXT TO one of the main stream takes the second granule set)
        void*& gran = h->trsv_alt ? h->d_trsv_gran2 : h->d_trsv_gran;
        long long& gep = h->trsv_alt ? h->trsv_gran2_epoch : h->trsv_gran_epoch;
        if (!gran) GPK_HIP(h, hipMalloc(&gran, (size_t)GPK_MAX_TRSV_BLOCKS * NB * sizeof(TrsvGran)));
        if (gep == 0) GPK_HIP(h, hipMemsetAsync(gran, 0, (size_t)GPK_MAX_TRSV_BLOCKS * NB * sizeof(TrsvGran), h->stream));
        const long long ep = ++gep;
        if (trans) trsv_gran_kernel<true><<<nblk, 256, 0, h->stream>>>(L, ldl, n, x, (TrsvGran*)gran, ep);
        else       trsv_gran_kernel<false><<<nblk, 256, 0, h->stream>>>(L, ldl, n, x, (TrsvGran*)gran, ep);
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    if (!trans) {
        for (int b = 0; b < nblk; ++b) {
            const int r0 = b * NB, nb = (n - r0 < NB) ? n - r0 : NB;
            trsv_diag_kernel<false><<<1, 64, 0, h->stream>>>(L + (long)r0 * ldl + r0, ldl, nb, x + r0);
            const int rest = n - r0 - nb;
            if (rest > 0)
                gemv_rows_kernel<<<gpk_ceil_div(rest, 4), 256, 0, h->stream>>>(L + (long)(r0 + nb) * ldl + r0, ldl, rest, nb, x + r0, x + r0 + nb);
        }
    } else {
        for (int b = nblk - 1; b >= 0; --b) {
            const int r0 = b * NB, nb = (n - r0 < NB) ? n - r0 : NB;
            trsv_diag_kernel<true><<<1, 64, 0, h->stream>>>(L + (long)r0 * ldl + r0, ldl, nb, x + r0);
            if (r0 > 0)
                gemv_cols_kernel<<<gpk_ceil_div(r0, 256), 256, 0, h->stream>>>(L + (long)r0 * ldl, ldl, nb, r0, x + r0, x);
        }
    }
    GPK_LAUNCH_CHECK(h);
    return 0;
}

int gpk_i_dot(gpk_handle h, const double* x, const double* y, int n, double* d_out) {
    dot_kernel<<<1, 1024, 0, h->stream>>>(x, y, n, d_out);
    GPK_LAUNCH_CHECK(h);
    return 0;
}

// ---- C ABI ------------------------------------------------------------------------------------------------------


extern "C" int gpk_potrf(gpk_handle h, double* A, int n, int lda, int* host_info) {
    if (!h || !A || n < 0 || lda < n) return GPK_ERR_ARG;
    GPK_HIP(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
    GPK_TRY(gpk_i_potrf(h, A, n, lda, 0));
    if (host_info) {
        GPK_HIP(h, hipMemcpyAsync(host_info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        GPK_HIP(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" int gpk_potrf_panel(gpk_handle h, double* A, int nrows, int ncols, int lda, int* host_info) {
    if (!h || !A || ncols < 0 || nrows < ncols || lda < ncols) return GPK_ERR_ARG;
    GPK_HIP(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
    GPK_TRY(gpk_i_potrf_panel(h, A, nrows, ncols, lda, 0));
    if (host_info) {
        GPK_HIP(h, hipMemcpyAsync(host_info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        GPK_HIP(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

// Panel step without any host traffic: the pivot status accumulates in the handle's device-side info word (first failure wins),
// indices offset by pivot_base; gpk_info_reset before the first panel of a factorisation, ONE gpk_info_read at its end.
extern "C" int gpk_potrf_panel_at(gpk_handle h, double* A, int nrows, int ncols, int lda, int pivot_base) {
    if (!h || !A || ncols < 0 || nrows < ncols || lda < ncols || pivot_base < 0) return GPK_ERR_ARG;
    return gpk_i_potrf_panel(h, A, nrows, ncols, lda, pivot_base);
}

extern "C" int gpk_info_reset(gpk_handle h) {
    if (!h) return GPK_ERR_ARG;
    GPK_HIP(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
    return 0;
}

extern "C" int gpk_info_read(gpk_handle h, int* host_info) {
    if (!h || !host_info) return GPK_ERR_ARG;
    GPK_HIP(h, hipMemcpyAsync(host_info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int gpk_trsm(gpk_handle h, int trans, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
    if (!h || !L || !B || n < 0 || nrhs < 0 || ldl < n || ldb < nrhs) return GPK_ERR_ARG;
    if (nrhs == 1 && ldb == 1) return gpk_i_trsv(h, trans != 0, L, n, ldl, B);
    return gpk_i_trsm_left_mt(h, trans != 0, L, n, ldl, B, nrhs, ldb);
}

extern "C" int gpk_trsm_lz(gpk_handle h, const double* L, int n, int ldl, double* B, int nrhs, int ldb, int lead) {
    if (!h || !L || !B || n < 0 || nrhs < 0 || ldl < n || ldb < nrhs) return GPK_ERR_ARG;
    if (lead <= 0) return gpk_i_trsm_left_mt(h, false, L, n, ldl, B, nrhs, ldb);
    return gpk_i_trsm_left_lz(h, L, n, ldl, B, nrhs, ldb, lead, 0);
}

extern "C" int gpk_trtri_diag(gpk_handle h, const double* L, int n, int ldl, double* Dinv, int block) {
    if (!h || !L || !Dinv || n < 0 || ldl < n) return GPK_ERR_ARG;
    return gpk_i_trtri_diag(h, L, n, ldl, Dinv, block);
}

extern "C" int gpk_trsm_dinv(gpk_handle h, const double* L, const double* Dinv, int block, int n, int ldl, double* B, int nrhs, int ldb,
                             double* X, int ldx, int lead) {
    if (!h || !L || !Dinv || !B || !X || n < 0 || nrhs < 0 || ldl < n || ldb < nrhs || ldx < nrhs) return GPK_ERR_ARG;
    if (!dinv_block_ok(block)) return gpk_bad_arg(h, "trsm_dinv: block size must be 256, 512, 1024 or 2048");
    if (B == X) return gpk_bad_arg(h, "trsm_dinv: X must not alias B");
    return gpk_i_trsm_left_dinv(h, L, Dinv, block, n, ldl, B, ldb, X, ldx, nrhs, lead > 0 ? lead : 0, 0);
}

extern "C" int gpk_trsm_right_lt(gpk_handle h, const double* L, int n, int ldl, double* X, int m, int ldx) {
    if (!h || !L || !X || n < 0 || m < 0 || ldl < n || ldx < n) return GPK_ERR_ARG;
    return gpk_i_trsm_right_lt(h, L, n, ldl, X, m, ldx);
}

extern "C" int gpk_potrs(gpk_handle h, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
    GPK_TRY(gpk_trsm(h, 0, L, n, ldl, B, nrhs, ldb));
    return gpk_trsm(h, 1, L, n, ldl, B, nrhs, ldb);
}

#ifdef GPK_DEV
#include "dev/gpk_factor_dev_abi.inc"    // gpk_debug_stamps, gpk_debug_overlap_probe (include/gpk_dev.h): development build only
#endif

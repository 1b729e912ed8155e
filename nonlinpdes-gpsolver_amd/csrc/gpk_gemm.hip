// gpk_gemm.hip -- fp64 GEMM/SYRK on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64).
//
// The one contraction kernel behind every O(n^3) step of the path: the SYRK H = S^T S of Hessian_GN
// (reference src/PDEs.py:100-102,295-307,453-455; src/InverseProblems.py:149-151), the trailing updates of the
// Cholesky factorisation (jnp.linalg.cholesky, src/PDEs.py:77,273,413) and the off-diagonal updates of the
// triangular solves (jnp.linalg.solve(self.L, .), src/PDEs.py:86,97,...).
//
// C[m,n] <- alpha * sum_k opA[m,k] opB[k,n] + beta * C[m,n], everything row-major fp64.
//   TA = false: A stored [m][k] (k contiguous)      TA = true: A stored [k][m] (m contiguous)
//   TB = false: B stored [k][n] (n contiguous)      TB = true: B stored [n][k] (k contiguous)
//
// Tiling for wave64 / MFMA 16x16x4 f64 (one instruction = 2048 flop, 64 cycles on a SIMD):
//   workgroup = 4 waves in a 2x2 grid, wave tile WM x WN = (WM/16) x (WN/16) MFMA accumulators (4 f64/lane each);
//   K is consumed in slabs of BK = 16 staged through LDS, double-buffered, global->register prefetch of slab t+1
//   issued before the MFMAs of slab t (one barrier per slab).
//   LDS images are padded so that every ds_read_b64 of a fragment is bank-conflict free:
//     k-contiguous operand : [row][BK+2]   (144-byte rows: 16 rows x 2 k's hit 32 distinct 8-byte bank pairs)
//     m-contiguous operand : [k][BM+16]    (consecutive k rows offset by 128 bytes mod 256)
//   MFMA operand map (cdna_hip_programming.md:247-251): A lane l = A[l&15][k=l>>4], B lane l = B[k=l>>4][l&15],
//   C/D lane l reg r = C[(l>>4)+4r][l&15].
// Workgroup -> tile map is XCD-aware: workgroups go round-robin over the 8 XCDs (block b on XCD (b + offset) % 8, the
// offset carried over from the previous launch: tools/xcc_probe.py reads HW_REG_XCC_ID), so blocks with equal b % 8 share
// an L2 and logical tile ids are handed out in
// 8 contiguous chunks (one per XCD L2), ordered in groups of 8 tile rows so that a chunk re-uses its A/B panels.
#include "gpk_common.h"

#include <algorithm>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int BK = 16;

struct SkSeg {
    int tm, tn;        // tile (a triangular-operand launch lists single row tiles, not pairs)
    int kbeg, kend;    // slabs
    int part, nparts;  // nparts == 1: the whole tile (direct epilogue); else partial `part` of `nparts` (K order)
    int slot0;         // first workspace slot (BM * BN doubles each) of the tile's partials
    int ticket;        // arrival counter of the tile
};

struct GemmArgs {
    int M, N, K;
    double alpha, beta;
    const double* A; long lda;
    const double* B; long ldb;
    double* C; long ldc;
    int lower_only;
    int lead_div;      // ... or, with a staircase of slope 1/lead_div, for k < (lead-1-n) / lead_div (gpk_ctx::lead_div)
    int lead;          // operand B (stored [k][n]) has column n zero for k < lead-1-n: the tile with columns [n0, n0+BN)
                       // gets no contribution from k < lead - (n0 + BN), so its K loop starts there.  (SYRK S^T S,
                       // lower tiles: the A tile's columns are further right, i.e. non-zero even earlier.)
    GpkStair stair;    // nseg > 0 (and lead > 0): the piecewise profile replaces the closed form -- the tile with columns [n0, n0+BN) starts at
    int stair_col0, stair_row0;   // row gpk_stair_min(stair, stair_col0 + n0, stair_col0 + min(n0+BN, N)) - stair_row0 (Darcy system, gpk_gn.hip)
    int skip_upper;    // C is a block column whose top square is a diagonal block of a symmetric matrix: tiles entirely above
                       // that diagonal (m0 + BM <= n0) are not computed (their content is never read)
    int stagger;       // experiment, see the kernel
    int rev_k;         // leading-zero operands: the K range of a tile is [k0(column), K) with a different k0 per column tile, so
                       // tiles that start together walk different slabs at any moment and share nothing in L2.  Walking K
                       // DOWNWARDS from the common end K aligns them: concurrently started tiles read the same slabs of A.
    int tri_a;         // operand A (TA = false, stored [m][k]) is lower triangular: row m has no entries at k > m, so the tile
                       // with rows [m0, m0+BM) stops its K loop at m0+BM (the explicit inverses of diagonal blocks, gpk_trsm_dinv)
    int vecA, vecB;
    int ntm, ntn, ntiles;
    int ntm_full;      // tri_a: number of row tiles of C (ntm counts PAIRS of them then)
    int band;          // > 0: leading-zero launches are ordered in bands of this many row tiles (see map_tile)
    int row_order;     // leading-zero launches without bands: 1 = row-major tile order, contiguous per XCD (see map_tile)
    int nsuper;        // > 0: supertile schedule of the lower-triangular, leading-zero (SYRK) launch, see map_tile
    int splitk;        // > 1: every tile's K range is cut into `splitk` chunks, one workgroup each (blockIdx = tile * splitk + chunk);
                       // the chunks leave their accumulators in `ws`, take a ticket in cnt[tile], and the LAST arriver adds them up in
                       // chunk order (so the result does not depend on who was last) and writes C.  For launches of few tiles with a
                       // long K loop (the 512-column products of the pipelined phase: one wave of 100-900 tiles with K up to 8400)
    double* ws;        // splitk: ntiles * splitk * BM * BN doubles
    unsigned* cnt;     // splitk: one arrival counter per tile, zero before the launch, reset to zero by the last arriver
    // Tile-list ("stream-K") launches: the host cuts the launch's slab iterations -- all tiles in the launch's order, each with ITS
    // K range (leading zeros, triangular operand) -- into equal shares, one per RESIDENT workgroup slot, and hands every workgroup its
    // list of segments.  A tile cut between workgroups is finished by the last arriver exactly like a split-K tile (partials in
    // `ws`, K order).  No tail of half-empty rounds, no long-tile / short-tile imbalance; the grid is one workgroup per slot.
    const SkSeg* sk_segs;
    const int* sk_off; // segments of block b: [sk_off[b], sk_off[b + 1])
};


// KC = true : operand stored [x][k] (k contiguous);  KC = false : stored [k][x] (x contiguous)
template <bool KC, int BX, int NT = 256>
__device__ __forceinline__ void load_tile(const double* __restrict__ P, long ld, int x0, int X, int k0, int K, bool vec,
                                          d2 (&r)[BX * BK / (2 * NT)], const int t) {
    // interior tiles (workgroup-uniform test -> scalar branch): unguarded 16-byte loads
    if (vec && x0 + BX <= X && k0 + BK <= K) {
#pragma unroll
        for (int i = 0; i < BX * BK / (2 * NT); ++i) {
            const int lin = t + NT * i;
            const double* ptr = KC ? P + (long)(x0 + lin / (BK / 2)) * ld + k0 + (lin % (BK / 2)) * 2
                                   : P + (long)(k0 + lin / (BX / 2)) * ld + x0 + (lin % (BX / 2)) * 2;
            r[i] = *reinterpret_cast<const d2*>(ptr);
        }
        return;
    }
    // edge tiles / unaligned operands: clamped addresses + selects (no divergent branches)
#pragma unroll
    for (int i = 0; i < BX * BK / (2 * NT); ++i) {
        const int lin = t + NT * i;
        int x, k;
        if (KC) { x = x0 + lin / (BK / 2); k = k0 + (lin % (BK / 2)) * 2; }
        else    { k = k0 + lin / (BX / 2); x = x0 + (lin % (BX / 2)) * 2; }
        const int xc0 = min(x, X - 1), kc0 = min(k, K - 1);
        const int xc1 = KC ? xc0 : min(x + 1, X - 1), kc1 = KC ? min(k + 1, K - 1) : kc0;
        const double e0 = KC ? P[(long)xc0 * ld + kc0] : P[(long)kc0 * ld + xc0];
        const double e1 = KC ? P[(long)xc1 * ld + kc1] : P[(long)kc1 * ld + xc1];
        const bool v0 = (x < X) && (k < K);
        const bool v1 = KC ? ((x < X) && (k + 1 < K)) : ((k < K) && (x + 1 < X));
        r[i].x = v0 ? e0 : 0.0;
        r[i].y = v1 ? e1 : 0.0;
    }
}

template <bool KC, int BX, int NT = 256>
__device__ __forceinline__ void store_tile(double* __restrict__ lds, const d2 (&r)[BX * BK / (2 * NT)], const int t) {
#pragma unroll
    for (int i = 0; i < BX * BK / (2 * NT); ++i) {
        const int lin = t + NT * i;
        int off;
        if (KC) off = (lin / (BK / 2)) * (BK + 2) + (lin % (BK / 2)) * 2;
        else    off = (lin / (BX / 2)) * (BX + 16) + (lin % (BX / 2)) * 2;
        *reinterpret_cast<d2*>(lds + off) = r[i];
    }
}

// element (x, k) of a staged tile
template <bool KC, int BX>
__device__ __forceinline__ double lds_at(const double* __restrict__ lds, int x, int k) {
    return KC ? lds[x * (BK + 2) + k] : lds[k * (BX + 16) + x];
}

constexpr int SG_H = 8, SG_W = 4;                                    // supertile: 8 tile rows x 4 tile columns

__device__ __forceinline__ bool map_tile(const GemmArgs& g, const int b, int& tm, int& tn) {
    if (g.nsuper > 0) {
        // OPTIONAL schedule (off by default, gpk_debug_set(6, 1)) for SYRK S^T S on lower tiles with leading zeros.  A
        // tile's K loop starts at a row that depends on its column block, so only tiles of the same column group can
        // walk K in step and share operand slabs in L2.  Supertiles of 8 x 4 tiles (all 32 co-resident on ONE XCD:
        // hardware places block b on XCD b % 8) start at a common K offset and share their 4 column and 8 row panels;
        // supertiles are handed out longest first, serpentine over the XCDs.  Measured at BASELINE config 2 (round 1):
        // L2 hit rate 6 % -> 54 %, fetched bytes per launch 11.2 GB -> 5.4 GB, but 1.76 ms against 1.65 ms for the
        // default single-tile longest-first round-robin: the operand stream is served by the 256 MB Infinity Cache
        // either way (S is 269 MB) and the static per-XCD partition costs more than the traffic it saves.
        const int xcd = b & 7, idx = b >> 3;
        const int q = idx / (SG_H * SG_W);
        int s = q * 8 + ((q & 1) ? 7 - xcd : xcd);                   // serpentine: evens out the longest-first ramp over the XCDs
        const int within = idx % (SG_H * SG_W);
        if (s >= g.nsuper) return false;
        const int T = g.ntm, ncg = (T + SG_W - 1) / SG_W;
        int cg = ncg - 1;
        for (; cg >= 0; --cg) {                                       // column groups, longest K first
            const int nrg = (T - cg * SG_W + SG_H - 1) / SG_H;
            if (s < nrg) break;
            s -= nrg;
        }
        tn = cg * SG_W + within % SG_W;
        tm = cg * SG_W + s * SG_H + within / SG_W;
        return tn < T && tm < T && tm >= tn;
    }
    // XCD-aware bijective remap (cdna_hip_programming.md §5 "XCD swizzle must be bijective")
    const int nwg = g.ntiles;
    const int xcd = b & 7, idx = b >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // leading-zero operands: tiles further right run a longer K loop; hand the long ones out first so that the short
    // ones fill the tail of the launch
    // (no per-XCD chunking then: contiguous chunks of a work ramp would load the XCDs unevenly; round-robin keeps them level)
    if (g.lead > 0) logical = nwg - 1 - b;
    if (g.lower_only) {
        if (g.lead > 0) logical = b;
        int i = (int)((sqrt(8.0 * (double)logical + 1.0) - 1.0) * 0.5);
        while ((long)(i + 1) * (i + 2) / 2 <= logical) ++i;
        while ((long)i * (i + 1) / 2 > logical) --i;
        if (g.lead > 0 && g.band > 0) {
            // large products: bands of `band` tile rows from the bottom; inside a band column-major from the longest column (first
            // the band's own triangle, then the full columns to its left).  The band's `band` column panels of S stay in the
            // Infinity Cache while the band sweeps the other panels, instead of every column sweeping all panels below it.
            const int T = g.ntm, R = g.band;
            int r1 = T, rem = b;
            for (;;) {                                                // (at most T / R iterations, scalar)
                const int r0b = max(r1 - R, 0);
                const int cnt = r1 * (r1 + 1) / 2 - r0b * (r0b + 1) / 2;
                if (rem < cnt || r0b == 0) break;
                rem -= cnt; r1 = r0b;
            }
            const int r0 = max(r1 - R, 0), rows = r1 - r0;
            const int tri = rows * (rows + 1) / 2;
            if (rem < tri) {                                          // triangle: column r1-1-c holds rows r1-1-c .. r1-1
                int c = (int)((sqrt(8.0 * (double)rem + 1.0) - 1.0) * 0.5);
                while ((c + 1) * (c + 2) / 2 <= rem) ++c;
                while (c * (c + 1) / 2 > rem) --c;
                tn = r1 - 1 - c; tm = tn + (rem - c * (c + 1) / 2);
            } else {
                const int q = rem - tri;
                tn = r0 - 1 - q / rows; tm = r0 + q % rows;
            }
        } else if (g.lead > 0) {
            // leading zeros: work depends on the column block only.  Column-major from the right-most (longest) column:
            // strictly longest-first, and the tiles of a column -- same K range, dispatched back to back, every 8th on
            // the same XCD -- walk their shared column panel in step (L2 reuse without a static per-XCD partition).
            tn = g.ntm - 1 - i; tm = tn + (logical - i * (i + 1) / 2);
        } else {
            tm = i; tn = logical - i * (i + 1) / 2;
        }
    } else if (g.lead > 0) {
        if (g.band > 0) {
            // tall launches: bands of `band` row tiles, column-major from the longest column INSIDE a band -- the band's rows of A
            // (band * BM * K * 8 bytes, kept below ~100 MB) stay in the 256 MB Infinity Cache while the band sweeps the columns,
            // instead of the whole of A being streamed from HBM once per column tile (north-star size: 157 times 856 MB)
            const int per = g.band * g.ntn;
            const int bd = b / per;
            const int r0 = bd * g.band;
            const int rows = min(g.band, g.ntm - r0);
            const int rem = b - bd * per;
            tn = g.ntn - 1 - rem / rows;
            tm = r0 + rem % rows;
        } else if (g.row_order) {
            // ROW-major, one contiguous run of tile rows per XCD (block b runs on XCD b % 8): the tiles (tm, all tn) of a row --
            // which read the same 64 columns of A over nearly the same K range -- run back to back on ONE XCD, so that panel comes
            // from memory once per row instead of once per tile; the few column panels of B (ntn x K x 64 doubles) are what is
            // re-read, and they fit the Infinity Cache.  For the 512-column products of the pipelined phase (ntn = 8, see
            // gpk_factor.hip): column-major order re-reads the 269 MB of S eight times per product.  EXPERIMENT, off by default: measured
            // slower (see h->tune.row_order).
            const int nwg2 = g.ntiles, xcd2 = b & 7, q2 = nwg2 >> 3, r2 = nwg2 & 7;
            const int l2 = (xcd2 < r2 ? xcd2 * (q2 + 1) : r2 * (q2 + 1) + (xcd2 - r2) * q2) + (b >> 3);
            tm = l2 / g.ntn;
            tn = g.ntn - 1 - (l2 - tm * g.ntn);
        } else {
            tn = g.ntn - 1 - b / g.ntm;                               // column-major from the longest column, as above
            tm = b % g.ntm;
        }
    } else {
        constexpr int GROUP = 8;                                     // (4 / 16 / 32 measured at the north-star size: no difference)
        const int per_group = GROUP * g.ntn;
        const int gid = logical / per_group;
        const int first = gid * GROUP;
        const int gsz = min(g.ntm - first, GROUP);
        const int rem = logical - gid * per_group;
        tm = first + rem % gsz;
        tn = rem / gsz;
    }
    return true;
}

template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool TRI = false, bool SK = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, ((BM / WM) * (BN / WN) >= 16 ? 1 : 2)) void gemm_f64_kernel(GemmArgs gk) {
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;                   // threads: 4 waves, or 8 for the 128 x 64 tile
    constexpr bool PF2 = (BM * BN <= 64 * 64) || NT >= 512;                     // prefetch depth 2 for the small-tile configuration
    constexpr int WAVES_N = BN / WN;
    static_assert(NT == 256 || NT == 512 || NT == 1024, "4, 8 or 16 waves per workgroup");
    constexpr int A_SZ = TA ? BK * (BM + 16) : BM * (BK + 2);
    constexpr int B_SZ = TB ? BN * (BK + 2) : BK * (BN + 16);
    __shared__ __attribute__((aligned(16))) double smem[2 * (A_SZ + B_SZ)];
    double* const As = smem;
    double* const Bs = smem + 2 * A_SZ;

    const int nsplit_launch = SK ? 1 : gk.splitk;
    const int bt = nsplit_launch > 1 ? (int)blockIdx.x / nsplit_launch : (int)blockIdx.x;
    const int chunk_launch = nsplit_launch > 1 ? (int)blockIdx.x - bt * nsplit_launch : 0;
    int tm_map = 0, tn_map = 0;
    if (!SK && !map_tile(gk, bt, tm_map, tn_map)) return;
    // Lower-triangular A (tri_a): the K loop of row tile t ends at (t+1) BM, so tile t costs ~(t+1) units.  The grid then has
    // one workgroup per PAIR of row tiles (t, ntm_full-1-t) -- constant work per workgroup whatever the dispatcher does (with
    // one tile per workgroup, the four tiles a CU received had the same t: 4..64 slabs against an average of 34).
    // (TRI is a template parameter and the tile body a lambda: as a runtime loop around the body it cost every instantiation
    // 30-50 VGPRs and a wave of occupancy)
    if (gk.stagger > 0) {
        // EXPERIMENT (gpk_debug_set key 15, off by default): de-phase the workgroups that share a CU.  A one-wave launch with a
        // long K loop (1008 tiles, K = 4352) runs in lock-step -- the four workgroups of a CU load, wait at their barrier and
        // compute at the same moments -- and reaches 56-59 TFLOP/s; with a start-time stagger of slot x 1024 cycles it reaches
        // 67 (tools/gemm_tail_probe.py; launches of >= 2 waves de-phase by themselves: 68-70 either way).  Inside the
        // Gauss-Newton step the one-wave launches have K <= 1024 and the stagger changes nothing (per-launch times within 1 %),
        // and neither do the one-wave, long-K product launches of the pipelined phase (sum of launches 2.70-2.74 ms either way), so it
        // stays off.  The wave slot id (HW_REG_HW_ID[3:0]) differs between the co-resident workgroups.
        const int slot = __builtin_amdgcn_s_getreg((4 << 11) | 4) & 0xf;    // size 4, offset 0, register 4 (HW_ID)
        for (int i = 0; i < (slot & 3) * gk.stagger; ++i) __builtin_amdgcn_s_sleep(8);
    }
    // sk_*: the segment of a tile-list launch (SK); otherwise the K range follows from the launch parameters
    // (tile-list launches call this in a loop: the launch arguments arrive as a per-iteration copy and the thread index through an
    // opaque move, so that nothing the body derives from them is kept in registers across the whole K loop of the NEXT segment --
    // hoisted, they cost every instantiation 20-36 VGPRs and a wave of occupancy)
    auto tile = [&](const GemmArgs& g, const int tm, const int tn, const int sk_kbeg, const int sk_kend, const int sk_part, const int sk_nparts,
                    const int sk_slot0, const int sk_ticket) __attribute__((always_inline)) {
    int tix = threadIdx.x;
    if (SK) asm volatile("" : "+v"(tix));
    const int lane = tix & 63, wave = tix >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int li = lane & 15, lk = lane >> 4;
    const int m0 = tm * BM;
    const int n0 = tn * BN;
    if (!SK && g.skip_upper && m0 + BM <= n0) return;
    const int nsplit = SK ? sk_nparts : nsplit_launch;
    const int chunk = SK ? sk_part : chunk_launch;

    d4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    const int Kt = TRI ? min(g.K, m0 + BM) : g.K;                     // lower-triangular A: nothing right of the tile's last row
    const int nk = (Kt + BK - 1) / BK;
    int kt0 = 0;
    if (g.lead > 0) {                                                 // both operands are zero above this row (tm >= tn)
        // supertile schedule: every tile of a supertile starts where its right-most column block does (<= 12 slabs of
        // zeros extra), so that the 32 tiles walk K in step and share operand slabs in L2
        const int nlast = g.nsuper > 0 ? min((tn / SG_W) * SG_W + SG_W - 1, g.ntn - 1) * BN : n0;
        const int z = g.lead - (nlast + BN);
        kt0 = z > 0 ? (z / g.lead_div) / BK : 0;
        if (g.stair.nseg > 0) {                                       // piecewise profile (wave-uniform: scalar ALU)
            const int fr = gpk_stair_min(g.stair, g.stair_col0 + n0, g.stair_col0 + min(n0 + BN, g.N)) - g.stair_row0;
            kt0 = fr > 0 ? fr / BK : 0;
        }
        if (kt0 > nk) kt0 = nk;
    }
    int kbeg = kt0, kend = nk;                                       // this workgroup's slabs
    if (SK) {
        kbeg = sk_kbeg; kend = sk_kend;
    } else if (nsplit > 1) {
        const int len = (nk - kt0 + nsplit - 1) / nsplit;
        kbeg = min(kt0 + chunk * len, nk);
        kend = min(kbeg + len, nk);
    }
    auto compute = [&](int buf) {
        const double* __restrict__ as = As + buf * A_SZ;
        const double* __restrict__ bs = Bs + buf * B_SZ;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = lds_at<!TA, BM>(as, wm0 + 16 * i + li, 4 * ks + lk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = lds_at<TB, BN>(bs, wn0 + 16 * j + li, 4 * ks + lk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    constexpr int NA = BM * BK / (2 * NT), NB = BN * BK / (2 * NT);
    int kdone = kbeg;                                                 // slabs [kbeg, kdone) are accumulated
    if (PF2) {
        // Interior tiles of aligned operands, full slabs only: two slabs in flight with STRAIGHT-LINE unguarded 16-byte
        // loads (slab index clamped instead of branched around), so that the compiler's s_waitcnt before the LDS store
        // of slab t+1 can leave the four loads of slab t+2 outstanding (vmcnt(4)); with the loads behind branches it
        // fell back to vmcnt(0), i.e. to one slab in flight.  Ablation on 8192^3 (tools/gemm_exp_probe.py): 62.5 TF/s
        // with the feed, 73.9 without -- the feed latency, not the MFMA pipe or LDS, is what is left to hide.
        // Rows of A / columns of B beyond the matrix edge only feed rows / columns of C that are never stored, so edge
        // tiles need no masking in m and n, only in-bounds addresses (clamped below; for an m-contiguous operand the last
        // 16-byte pair may read the padding element at column X, which exists because the leading dimension is even).
        const bool fast = g.vecA && g.vecB;
        const int nkf = fast ? min(Kt / BK, kend) : kbeg;             // full slabs
        if (nkf > kbeg) {
            const int t = tix;
            const double* pa[NA]; const double* pb[NB];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int lin = t + NT * i;
                pa[i] = !TA ? g.A + (long)min(m0 + lin / (BK / 2), g.M - 1) * g.lda + (lin % (BK / 2)) * 2
                            : g.A + (long)(lin / (BM / 2)) * g.lda + min(m0 + (lin % (BM / 2)) * 2, (g.M - 1) & ~1);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int lin = t + NT * i;
                pb[i] = TB ? g.B + (long)min(n0 + lin / (BK / 2), g.N - 1) * g.ldb + (lin % (BK / 2)) * 2
                           : g.B + (long)(lin / (BN / 2)) * g.ldb + min(n0 + (lin % (BN / 2)) * 2, (g.N - 1) & ~1);
            }
            const long sa = !TA ? (long)BK : (long)BK * g.lda, sb = TB ? (long)BK : (long)BK * g.ldb;
            d2 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
            auto load = [&](int kt, d2 (&ra)[NA], d2 (&rb)[NB]) {
                int kc = min(kt, nkf - 1);                            // past the end: re-read the last slab (never used)
                if (g.rev_k) kc = (nkf - 1) - (kc - kbeg);            // walk K downwards (see GemmArgs::rev_k)
#pragma unroll
                for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const d2*>(pa[i] + kc * sa);
#pragma unroll
                for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const d2*>(pb[i] + kc * sb);
            };
            auto store = [&](int buf, const d2 (&ra)[NA], const d2 (&rb)[NB]) {
                store_tile<!TA, BM, NT>(As + buf * A_SZ, ra, tix);
                store_tile<TB, BN, NT>(Bs + buf * B_SZ, rb, tix);
            };
            load(kbeg, ra0, rb0);
            load(kbeg + 1, ra1, rb1);
            store(kbeg & 1, ra0, rb0);
            __syncthreads();
            for (int kt = kbeg; kt < nkf; kt += 2) {
                load(kt + 2, ra0, rb0);                               // registers 0 are free (slab kt sits in LDS)
                compute(kt & 1);
                store((kt + 1) & 1, ra1, rb1);
                __syncthreads();
                if (kt + 1 >= nkf) break;
                load(kt + 3, ra1, rb1);
                compute((kt + 1) & 1);
                store(kt & 1, ra0, rb0);
                __syncthreads();
            }
            kdone = nkf;
        }
    }
    if (kdone < kend) {                                               // edge tiles, unaligned operands, the partial slab
        d2 ra[NA], rb[NB];
        load_tile<!TA, BM, NT>(g.A, g.lda, m0, g.M, kdone * BK, Kt, g.vecA, ra, tix);
        load_tile<TB, BN, NT>(g.B, g.ldb, n0, g.N, kdone * BK, Kt, g.vecB, rb, tix);
        store_tile<!TA, BM, NT>(As + (kdone & 1) * A_SZ, ra, tix);
        store_tile<TB, BN, NT>(Bs + (kdone & 1) * B_SZ, rb, tix);
        __syncthreads();
        for (int kt = kdone; kt < kend; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < kend) {
                load_tile<!TA, BM, NT>(g.A, g.lda, m0, g.M, (kt + 1) * BK, Kt, g.vecA, ra, tix);
                load_tile<TB, BN, NT>(g.B, g.ldb, n0, g.N, (kt + 1) * BK, Kt, g.vecB, rb, tix);
            }
            compute(cur);
            if (kt + 1 < kend) {
                store_tile<!TA, BM, NT>(As + (cur ^ 1) * A_SZ, ra, tix);
                store_tile<TB, BN, NT>(Bs + (cur ^ 1) * B_SZ, rb, tix);
            }
            __syncthreads();
        }
    }

    if (nsplit > 1) {
        // Hand-over across workgroups (other CUs, other XCDs: private L1s, non-coherent L2s): the accumulators leave as
        // write-through (agent-scope, sc1) stores, every wave waits for the acknowledgements of its own, then one lane takes the
        // ticket; the last arriver reads the chunks with sc1 loads (MI355X_MICROARCH.md, "splitk-seam").
        const long tid = SK ? (long)sk_ticket : (long)tm * g.ntn + tn;
        double* const wt = g.ws + (SK ? (long)sk_slot0 : tid * nsplit) * (BM * BN);
        double* const mine = wt + (long)chunk * (BM * BN);
        // (16-byte write-through stores: two per accumulator tile and lane; as 8-byte agent-scope atomics they cost 2.7x per byte)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                double* dst = mine + ((i * TN + j) * NT + tix) * 4;
                const d2 lo = (d2){acc[i][j][0], acc[i][j][1]}, hi = (d2){acc[i][j][2], acc[i][j][3]};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" :: "v"(dst), "v"(lo), "v"(hi) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (the ticket travels through the first word of the operand buffers, which nobody reads any more: a variable of its own
        // would cost the 64x64 TN instantiation its fourth workgroup per CU -- 4 x 40 KB are exactly the 160 KB of LDS)
        unsigned* const s_ticket = reinterpret_cast<unsigned*>(smem);
        if (tix == 0) *s_ticket = __hip_atomic_fetch_add(g.cnt + tid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *s_ticket;
        if (SK) __syncthreads();                                      // (the next segment's first slab overwrites the ticket word)
        if (ticket != (unsigned)(nsplit - 1)) return;
        if (tix == 0) __hip_atomic_store(g.cnt + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        d4 sum[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) sum[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
        // (compiler-visible 8-byte agent-scope loads: loads issued from inline asm are not tracked, and with the large tile
        // configurations the compiler spilled their destination registers before the data had arrived)
        for (int c = 0; c < nsplit; ++c) {
            const double* src = wt + (long)c * (BM * BN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sum[i][j][r] += __hip_atomic_load(src + ((i * TN + j) * NT + tix) * 4 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = sum[i][j];
    }
    const bool has_beta = (g.beta != 0.0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm0 + 16 * i + lk + 4 * r;
            if (row >= g.M) continue;
            double* crow = g.C + (long)row * g.ldc;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn0 + 16 * j + li;
                if (col < g.N) {
                    double v = g.alpha * acc[i][j][r];
                    if (has_beta) v += g.beta * crow[col];
                    crow[col] = v;
                }
            }
        }
    }
    };
    if (SK) {
        const int s0 = gk.sk_off[blockIdx.x], s1 = gk.sk_off[blockIdx.x + 1];
        for (int si = s0; si < s1; ++si) {
            // the launch arguments re-read from the kernarg segment behind an opaque move of its address (scalar loads, a few dozen
            // dwords per segment): as loop invariants they would occupy ~40 SGPRs for the whole kernel and spill into VGPR lanes
            typedef const __attribute__((address_space(4))) GemmArgs* karg_ptr;
            karg_ptr gp = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(gp));
            GemmArgs g;
#define GPK_CP(f) g.f = gp->f;
            GPK_CP(M) GPK_CP(N) GPK_CP(K) GPK_CP(alpha) GPK_CP(beta) GPK_CP(A) GPK_CP(lda) GPK_CP(B) GPK_CP(ldb) GPK_CP(C) GPK_CP(ldc)
            GPK_CP(lower_only) GPK_CP(lead_div) GPK_CP(lead) GPK_CP(skip_upper) GPK_CP(stagger) GPK_CP(rev_k) GPK_CP(tri_a) GPK_CP(vecA) GPK_CP(vecB)
            GPK_CP(ntm) GPK_CP(ntn) GPK_CP(ntiles) GPK_CP(ntm_full) GPK_CP(band) GPK_CP(nsuper) GPK_CP(splitk) GPK_CP(ws) GPK_CP(cnt)
            GPK_CP(sk_segs) GPK_CP(sk_off) GPK_CP(row_order)
#undef GPK_CP
            // (wave-uniform by construction: through SGPRs, so that the loop costs no vector registers)
            const int* q = reinterpret_cast<const int*>(g.sk_segs + si);
            const int a0 = __builtin_amdgcn_readfirstlane(q[0]), a1 = __builtin_amdgcn_readfirstlane(q[1]);
            const int a2 = __builtin_amdgcn_readfirstlane(q[2]), a3 = __builtin_amdgcn_readfirstlane(q[3]);
            const int a4 = __builtin_amdgcn_readfirstlane(q[4]), a5 = __builtin_amdgcn_readfirstlane(q[5]);
            const int a6 = __builtin_amdgcn_readfirstlane(q[6]), a7 = __builtin_amdgcn_readfirstlane(q[7]);
            tile(g, a0, a1, a2, a3, a4, a5, a6, a7);
        }
    } else if (TRI) {
        const int hi = gk.ntm_full - 1 - tm_map;                       // the long tile first
        tile(gk, hi, tn_map, 0, 0, 0, 1, 0, 0);
        if (hi != tm_map) tile(gk, tm_map, tn_map, 0, 0, 0, 1, 0, 0);     // odd count: the middle tile is its own partner
    } else {
        tile(gk, tm_map, tn_map, 0, 0, 0, 1, 0, 0);
    }
}

// ---- rank-<=64 updates: everything in flight at once -----------------------------------------------------------------
// The K = 64 updates of the TRSM recursion and of the POTRF panels (C -= A B, ~130 launches per Gauss-Newton step at
// C2) spend their time in dependent memory round trips, not in MFMAs: with the slab-by-slab pipeline above a 64x64
// tile pays one ~3 us round trip per 16-deep slab plus one for the read-modify-write of C.  Here all four slabs of A
// and B and the C tile are requested before anything is waited for: one round trip, one barrier, 64 MFMAs per wave.
template <bool TA, bool TB, int BM>
__global__ __launch_bounds__(256, 2) void gemm_k64_kernel(GemmArgs g) {
    constexpr int BN = 64, WM = BM / 2, WN = 32, TM = WM / 16, TN = 2, NK = 4;
    constexpr int A_SZ = TA ? BK * (BM + 16) : BM * (BK + 2);
    constexpr int B_SZ = TB ? BN * (BK + 2) : BK * (BN + 16);
    __shared__ __attribute__((aligned(16))) double smem[NK * (A_SZ + B_SZ)];
    double* const As = smem;
    double* const Bs = smem + NK * A_SZ;
    __builtin_amdgcn_s_setprio(3);                                   // part of a latency chain (panel loop): see potrf_panel_kernel
    int tm, tn;
    if (!map_tile(g, blockIdx.x, tm, tn)) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int li = lane & 15, lk = lane >> 4;

    d2 ra[NK][BM * BK / 512], rb[NK][BN * BK / 512];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        load_tile<!TA, BM>(g.A, g.lda, m0, g.M, kt * BK, g.K, g.vecA, ra[kt], (int)threadIdx.x);
        load_tile<TB, BN>(g.B, g.ldb, n0, g.N, kt * BK, g.K, g.vecB, rb[kt], (int)threadIdx.x);
    }
    // accumulators start from C (sign folded in: the only combinations routed here are beta = 0, or beta = 1 with
    // alpha = +-1, for which acc = C/alpha is exact)
    d4 acc[TM][TN];
    const bool has_c = (g.beta != 0.0);
    const double cs = has_c ? g.beta / g.alpha : 0.0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = min(m0 + wm0 + 16 * i + lk + 4 * r, g.M - 1);
                const int col = min(n0 + wn0 + 16 * j + li, g.N - 1);
                acc[i][j][r] = has_c ? cs * g.C[(long)row * g.ldc + col] : 0.0;
            }
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        store_tile<!TA, BM>(As + kt * A_SZ, ra[kt], (int)threadIdx.x);
        store_tile<TB, BN>(Bs + kt * B_SZ, rb[kt], (int)threadIdx.x);
    }
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        const double* __restrict__ as = As + kt * A_SZ;
        const double* __restrict__ bs = Bs + kt * B_SZ;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = lds_at<!TA, BM>(as, wm0 + 16 * i + li, 4 * ks + lk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = lds_at<TB, BN>(bs, wn0 + 16 * j + li, 4 * ks + lk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm0 + 16 * i + lk + 4 * r;
            if (row >= g.M) continue;
            double* crow = g.C + (long)row * g.ldc;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn0 + 16 * j + li;
                if (col < g.N) crow[col] = g.alpha * acc[i][j][r];
            }
        }
}

// Flops a launch EXECUTES, from the K ranges the kernel derives for its tiles (leading zeros, triangular operand, skipped upper
// tiles; rows and columns past the edge of C not counted): accumulated per phase of the Gauss-Newton step while the per-phase
// timing is on (gpk_prof_enable) and read back with gpk_prof_read_flops -- the numerator of bench.py's roofline legs, taken from
// the launch logic itself instead of a host-side model of it.
void prof_count(gpk_handle h, const GemmArgs& g, int BM, int BN) {
    if (!h->prof) return;
    const int ph = h->prof_phase < 0 ? 0 : h->prof_phase > 3 ? 3 : h->prof_phase;
    double fl = 0.0;
    const int ntm_full = g.tri_a ? g.ntm_full : g.ntm;
    for (int tn = 0; tn < g.ntn; ++tn) {
        const int n0 = tn * BN, bn = std::min(BN, g.N - n0);
        int k0 = 0;
        if (g.lead > 0) {
            const int z = g.lead - (n0 + BN);
            k0 = z > 0 ? ((z / g.lead_div) / BK) * BK : 0;
            if (g.stair.nseg > 0) {
                const int fr = gpk_stair_min(g.stair, g.stair_col0 + n0, g.stair_col0 + std::min(n0 + BN, g.N)) - g.stair_row0;
                k0 = fr > 0 ? (fr / BK) * BK : 0;
            }
        }
        if (g.tri_a) {
            for (int tm = 0; tm < ntm_full; ++tm) {
                const int m0 = tm * BM, bm = std::min(BM, g.M - m0), kt = std::min(g.K, m0 + BM);
                if (kt > k0) fl += 2.0 * bm * bn * (double)(kt - k0);
            }
            continue;
        }
        int mfirst = 0;                                              // first row that belongs to a computed tile of this column
        if (g.lower_only) mfirst = n0;                               // (square tiles: row tile tn)
        else if (g.skip_upper) mfirst = (n0 / BM) * BM;              // tiles with m0 + BM <= n0 are skipped
        if (g.M > mfirst && g.K > k0) fl += 2.0 * (double)(g.M - mfirst) * bn * (double)(g.K - k0);
    }
    h->prof_flops[ph] += fl;
    h->prof_launches[ph] += 1;
}

template <int BM>
int launch_k64_bm(gpk_handle h, bool ta, bool tb, GemmArgs& g) {
    if (h->prof) {
        const int ph = h->prof_phase < 0 ? 0 : h->prof_phase > 3 ? 3 : h->prof_phase;
        h->prof_flops[ph] += 2.0 * g.M * (double)g.N * g.K;
        h->prof_launches[ph] += 1;
    }
    g.ntm = gpk_ceil_div(g.M, BM);
    g.ntn = gpk_ceil_div(g.N, 64);
    g.ntiles = g.ntm * g.ntn;
    g.nsuper = 0;
    dim3 grid(g.ntiles), block(256);
    if (!ta && !tb) gemm_k64_kernel<false, false, BM><<<grid, block, 0, h->stream>>>(g);
    else if (!ta && tb) gemm_k64_kernel<false, true, BM><<<grid, block, 0, h->stream>>>(g);
    else if (ta && !tb) gemm_k64_kernel<true, false, BM><<<grid, block, 0, h->stream>>>(g);
    else gemm_k64_kernel<true, true, BM><<<grid, block, 0, h->stream>>>(g);
    GPK_LAUNCH_CHECK(h);
    return 0;
}

int launch_k64(gpk_handle h, bool ta, bool tb, GemmArgs& g) {
    // a rank-64 update is one round trip to memory plus 64 MFMAs per wave of a 64 x 64 tile: with fewer than ~4 tiles per
    // CU, halving the tile spreads the MFMA phase over twice as many workgroups
    const long t64 = (long)gpk_ceil_div(g.M, 64) * gpk_ceil_div(g.N, 64);
    if (h->tune.k64_small && t64 < 4 * h->num_cu && g.M > 64) return launch_k64_bm<32>(h, ta, tb, g);
    return launch_k64_bm<64>(h, ta, tb, g);
}


// ---- tile-list ("stream-K") plans: built on the host per launch SHAPE, cached on the device ------------------------------------

struct SkKey {
    int bm, bn, M, N, K, lower, lead, lead_div, tri, skip_upper, band, G;
    bool operator==(const SkKey& o) const { return memcmp(this, &o, sizeof(SkKey)) == 0; }
};
struct SkPlan {
    SkKey key;
    std::vector<SkSeg> segs;                                         // (host copies stay alive: the upload may read them late)
    std::vector<int> off;
    SkSeg* d_segs = nullptr;
    int* d_off = nullptr;
    int nblocks = 0, nslots = 0, ntickets = 0;
    unsigned long stamp = 0;
};
struct SkCache {
    std::vector<SkPlan*> plans;
    char* arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    unsigned long clock = 0;
};

// the tiles of a launch in the order the one-tile-per-workgroup kernel would take them (see map_tile), with their slab ranges
struct SkTile { int tm, tn, k0, k1; };
template <int BM, int BN>
void sk_enumerate(const GemmArgs& g, std::vector<SkTile>& out) {
    out.clear();
    const int T = g.tri_a ? g.ntm_full : g.ntm;                      // row tiles
    auto push = [&](int tm, int tn) {
        const int m0 = tm * BM, n0 = tn * BN;
        if (g.skip_upper && m0 + BM <= n0) return;
        const int Kt = g.tri_a ? std::min(g.K, m0 + BM) : g.K;
        const int nk = (Kt + BK - 1) / BK;
        int kt0 = 0;
        if (g.lead > 0) {
            const int z = g.lead - (n0 + BN);
            kt0 = z > 0 ? (z / g.lead_div) / BK : 0;
            if (kt0 > nk) kt0 = nk;
        }                                                            // (launches with a piecewise profile never take tile lists: launch_cfg)
        out.push_back({tm, tn, kt0, nk});
    };
    if (g.lower_only) {
        if (g.lead > 0 && g.band > 0) {
            for (int r1 = T; r1 > 0;) {                               // bands of tile rows from the bottom
                const int r0 = std::max(r1 - g.band, 0), rows = r1 - r0;
                for (int c = 0; c < rows; ++c) for (int tm = r1 - 1 - c; tm < r1; ++tm) push(tm, r1 - 1 - c);
                for (int tn = r0 - 1; tn >= 0; --tn) for (int tm = r0; tm < r1; ++tm) push(tm, tn);
                r1 = r0;
            }
        } else if (g.lead > 0) {
            for (int tn = T - 1; tn >= 0; --tn) for (int tm = tn; tm < T; ++tm) push(tm, tn);
        } else {
            for (int tm = 0; tm < T; ++tm) for (int tn = 0; tn <= tm; ++tn) push(tm, tn);
        }
        return;
    }
    auto rows_of = [&](int r0, int r1, int tn) {                      // row tiles [r0, r1) of one column; triangular operand: long tiles first
        if (g.tri_a) { for (int tm = r1 - 1; tm >= r0; --tm) push(tm, tn); }
        else         { for (int tm = r0; tm < r1; ++tm) push(tm, tn); }
    };
    if (g.lead > 0) {
        const int band = (g.band > 0 && !g.tri_a) ? g.band : T;
        for (int r0 = 0; r0 < T; r0 += band)
            for (int tn = g.ntn - 1; tn >= 0; --tn) rows_of(r0, std::min(r0 + band, T), tn);
    } else {
        for (int r0 = 0; r0 < T; r0 += 8)
            for (int tn = 0; tn < g.ntn; ++tn) rows_of(r0, std::min(r0 + 8, T), tn);
    }
}

template <int BM, int BN>
SkPlan* sk_build(gpk_handle h, const GemmArgs& g, int G, const SkKey& key) {
    std::vector<SkTile> tiles;
    sk_enumerate<BM, BN>(g, tiles);
    // Row tiles stay with "their" XCD, as in the one-tile-per-workgroup order (consecutive blocks = consecutive row tiles of a column,
    // block b on XCD b % 8): the shares are handed to the XCDs in contiguous runs (below), so the tile sequence is regrouped by
    // tm mod 8 first -- every XCD then works on 1/8 of the rows of A, which stay in ITS L2 while it sweeps the columns.  (Without
    // this every XCD streamed all of A: the update launches of the solve phase ran 1.3 - 2.5x slower than one tile per workgroup.)
    if (h->tune.sk_rowclass) std::stable_sort(tiles.begin(), tiles.end(), [](const SkTile& a, const SkTile& b) { return (a.tm & 7) < (b.tm & 7); });
    long total = 0;
    for (const SkTile& t : tiles) total += t.k1 - t.k0;
    if (tiles.empty() || total <= 0) return nullptr;
    SkPlan* p = new SkPlan();
    p->key = key;
    G = (int)std::min<long>(G, std::max<long>(1, total / 8));        // at least 8 slabs per workgroup
    G = std::max(8, (G / 8) * 8);
    // share boundaries in units of slabs over the concatenated tiles, snapped to nearby tile boundaries
    std::vector<long> pre(tiles.size() + 1, 0);
    for (size_t i = 0; i < tiles.size(); ++i) pre[i + 1] = pre[i] + (tiles[i].k1 - tiles[i].k0);
    std::vector<long> bnd((size_t)G + 1);
    size_t ti = 0;
    for (int w = 0; w <= G; ++w) {
        long b = (long)((double)total * w / G + 0.5);
        if (w == 0) b = 0;
        if (w == G) b = total;
        while (ti + 1 < pre.size() && pre[ti + 1] <= b) ++ti;        // pre[ti] <= b < pre[ti + 1] (or the end)
        if (b - pre[ti] < h->tune.sk_snap) b = pre[ti];
        else if (ti + 1 < pre.size() && pre[ti + 1] - b < h->tune.sk_snap) b = pre[ti + 1];
        if (w > 0 && b < bnd[w - 1]) b = bnd[w - 1];
        bnd[w] = b;
    }
    // parts per tile
    std::vector<int> nparts(tiles.size(), 0), slot0(tiles.size(), 0), seen(tiles.size(), 0);
    {
        size_t t = 0;
        for (int w = 0; w < G; ++w) {
            long a = bnd[w];
            const long e = bnd[w + 1];
            while (a < e) {
                while (pre[t + 1] <= a) ++t;
                const long stop = std::min(e, pre[t + 1]);
                nparts[t] += 1;
                a = stop;
            }
        }
    }
    int slots = 0, tickets = 0;
    std::vector<int> ticket(tiles.size(), 0);
    for (size_t t = 0; t < tiles.size(); ++t) {
        if (nparts[t] > 1) { slot0[t] = slots; slots += nparts[t]; ticket[t] = tickets++; }
    }
    // segments per share; zero-length tiles (all-zero K range) ride with the share that owns their position
    std::vector<std::vector<SkSeg>> per((size_t)G);
    {
        size_t t = 0;
        for (int w = 0; w < G; ++w) {
            long a = bnd[w];
            const long e = bnd[w + 1];
            while (a < e) {
                while (pre[t + 1] <= a) ++t;
                const long stop = std::min(e, pre[t + 1]);
                const SkTile& tl = tiles[t];
                per[w].push_back({tl.tm, tl.tn, (int)(tl.k0 + (a - pre[t])), (int)(tl.k0 + (stop - pre[t])), seen[t], nparts[t], slot0[t], ticket[t]});
                seen[t] += 1;
                a = stop;
            }
        }
        int w = 0;
        for (size_t i = 0; i < tiles.size(); ++i) {
            if (tiles[i].k1 > tiles[i].k0) continue;
            while (w + 1 < G && bnd[w + 1] <= pre[i] && bnd[w + 1] < total) ++w;
            per[w].push_back({tiles[i].tm, tiles[i].tn, tiles[i].k0, tiles[i].k0, 0, 1, 0, 0});
        }
    }
    // hardware block b runs on XCD b % 8: consecutive shares go to ONE XCD (its L2 sees neighbouring tiles)
    p->off.assign((size_t)G + 1, 0);
    p->segs.clear();
    const int per_xcd = G / 8;
    for (int b = 0; b < G; ++b) {
        const int w = (b % 8) * per_xcd + b / 8;
        p->off[b] = (int)p->segs.size();
        p->segs.insert(p->segs.end(), per[w].begin(), per[w].end());
    }
    p->off[G] = (int)p->segs.size();
    p->nblocks = G; p->nslots = slots; p->ntickets = tickets;
    return p;
}

int sk_upload(gpk_handle h, SkCache* c, SkPlan* p) {
    const size_t need = ((p->segs.size() * sizeof(SkSeg) + 255) & ~(size_t)255) + ((p->off.size() * sizeof(int) + 255) & ~(size_t)255);
    if (!c->arena) {
        c->arena_cap = (size_t)48 << 20;
        GPK_HIP(h, hipMalloc((void**)&c->arena, c->arena_cap));
    }
    if (c->arena_used + need > c->arena_cap) {                       // full: drop every plan (rare: hundreds of distinct launch shapes)
        GPK_HIP(h, hipDeviceSynchronize());
        for (SkPlan* q : c->plans) if (q != p) delete q;
        c->plans.clear();
        c->arena_used = 0;
        if (need > c->arena_cap) return 1;
    }
    p->d_segs = (SkSeg*)(c->arena + c->arena_used);
    p->d_off = (int*)(c->arena + c->arena_used + ((p->segs.size() * sizeof(SkSeg) + 255) & ~(size_t)255));
    c->arena_used += need;
    GPK_HIP(h, hipMemcpyAsync(p->d_segs, p->segs.data(), p->segs.size() * sizeof(SkSeg), hipMemcpyHostToDevice, h->stream));
    GPK_HIP(h, hipMemcpyAsync(p->d_off, p->off.data(), p->off.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    // (tables are read by launches on other streams of this handle as well: make them visible to all of them)
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// plan for this launch, or nullptr when the launch should take the one-tile-per-workgroup path
template <int BM, int BN>
SkPlan* sk_plan_for(gpk_handle h, const GemmArgs& g, int G) {
    if (!h->sk_cache) h->sk_cache = new SkCache();
    SkCache* c = (SkCache*)h->sk_cache;
    SkKey key;
    memset(&key, 0, sizeof key);
    key.bm = BM; key.bn = BN; key.M = g.M; key.N = g.N; key.K = g.K; key.lower = g.lower_only; key.lead = g.lead; key.lead_div = g.lead_div;
    key.tri = g.tri_a; key.skip_upper = g.skip_upper; key.band = g.band; key.G = G;
    for (SkPlan* p : c->plans) if (p->key == key) { p->stamp = ++c->clock; return p->nblocks > 0 ? p : nullptr; }
    SkPlan* p = sk_build<BM, BN>(h, g, G, key);
    if (!p) { p = new SkPlan(); p->key = key; }                       // remembered as "not worth it"
    else if ((size_t)p->nslots * BM * BN * sizeof(double) > h->splitk_ws_cap || p->ntickets > h->splitk_cnt_cap || sk_upload(h, c, p) != 0) {
        p->nblocks = 0;
        (void)hipGetLastError();
    }
    p->stamp = ++c->clock;
    c->plans.push_back(p);
    return p->nblocks > 0 ? p : nullptr;
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(gpk_handle h, bool ta, bool tb, GemmArgs& g) {
    g.ntm = gpk_ceil_div(g.M, BM);
    g.ntm_full = g.ntm;
    if (g.tri_a) g.ntm = (g.ntm + 1) / 2;                            // one workgroup per pair of row tiles (see the kernel)
    g.ntn = gpk_ceil_div(g.N, BN);
    g.ntiles = g.lower_only ? g.ntm * (g.ntm + 1) / 2 : g.ntm * g.ntn;
    g.nsuper = 0;
    prof_count(h, g, BM, BN);
    int nblocks = g.ntiles;
    if (g.lower_only && g.lead > 0 && h->tune.supertile && g.lead_div == 1 && g.stair.nseg == 0) {
        const int T = g.ntm, ncg = gpk_ceil_div(T, SG_W);
        for (int cg = 0; cg < ncg; ++cg) g.nsuper += gpk_ceil_div(T - cg * SG_W, SG_H);
        nblocks = 8 * gpk_ceil_div(g.nsuper, 8) * SG_H * SG_W;
    }
    g.band = 0;
    g.row_order = (h->tune.row_order && g.lead > 0 && !g.lower_only && !g.tri_a && g.skip_upper && g.ntn <= 16) ? 1 : 0;
    if (g.lead > 0 && g.lower_only && h->tune.syrk_band > 0) {
        const double panel = (double)BM * g.K * sizeof(double);       // one column panel of S
        if ((double)g.ntm * panel > 4.0 * h->tune.syrk_band * 1048576.0) {
            int r = (int)(h->tune.syrk_band * 1048576.0 / panel);
            g.band = r < 2 ? 2 : r;
        }
    }
    if (g.lead > 0 && !g.lower_only && !g.tri_a && h->tune.band_mb > 0) {
        const double panel = (double)BM * g.K * sizeof(double);       // one row tile's rows of A
        if ((double)g.ntm * panel > 2.0 * h->tune.band_mb * 1048576.0) {
            int r = (int)(h->tune.band_mb * 1048576.0 / panel);
            g.band = r < 1 ? 1 : r;
        }
    }
    g.splitk = 1; g.ws = nullptr; g.cnt = nullptr;
    int want = h->splitk_req;
    if (want <= 1 && h->tune.force_splitk > 1 && gpk_i_splitk_reserve(h) == 0) want = h->tune.force_splitk;
    if (want > 1 && !g.lower_only && !g.tri_a && h->d_splitk_ws && h->d_splitk_cnt) {
        int s = want > 16 ? 16 : want;
        while (s > 1 && (g.K / BK) / s < 16) --s;                    // chunks of at least 16 slabs
        while (s > 1 && (size_t)g.ntiles * s * BM * BN * sizeof(double) > h->splitk_ws_cap) --s;
        if (s > 1 && g.ntiles <= h->splitk_cnt_cap) {
            g.splitk = s; g.ws = h->d_splitk_ws; g.cnt = h->d_splitk_cnt;
            nblocks = g.ntiles * s;
        }
    }
    dim3 grid(nblocks), block((BM / WM) * (BN / WN) * 64);
    const size_t dyn = (size_t)h->tune.gemm_extra_lds;
    g.sk_segs = nullptr; g.sk_off = nullptr;
    // Tile-list launch?  Resident workgroup slots of this configuration (waves per workgroup -> workgroups per CU: 4 waves 4 (5 for the
    // 32-row tile), 8 waves 2, 16 waves 1); worth it when the launch is only a few rounds of them -- then the last, partly filled
    // round and the spread of tile lengths (leading zeros, triangular operand) cost a large share of its time.
    if (h->tune.sk && g.splitk == 1 && g.nsuper == 0 && !h->no_sk && !g.rev_k && g.K >= 4 * BK && h->num_cu >= 8 && g.stair.nseg == 0) {
        constexpr int WAVES = (BM / WM) * (BN / WN);
        const int per_cu = WAVES >= 16 ? 1 : WAVES >= 8 ? 2 : (BM == 32 ? 5 : 4);
        const int G = ((h->num_cu * per_cu) / 8) * 8;
        const long eff = g.tri_a ? 2L * g.ntiles : (long)g.ntiles;
        // Automatic mode: plain operands only.  Measured on the launches of the Gauss-Newton step (config 2, tools/sk_trace.sh): the
        // leading-zero updates and the triangular products of the solve phase already run at 72-88 % of the clock-limited rate with
        // one tile per workgroup (longest tile first; triangular tiles paired to constant work), and cutting their tiles costs more
        // in partial sums than the balance returns (179 -> 250 us, 566 -> 643, 1255 -> 1208 for the three updates; 107-287 -> 119-339
        // for the triangular products).  On plain products whose tile count is not a whole number of rounds the lists gain 12-21 %
        // (tools/sk_probe.py: 1056 tiles 0.375 -> 0.329 ms, 2080 tiles 0.69 -> 0.61 ms), at whole rounds they are neutral.
        const bool plain = g.lead == 0 && !g.tri_a && !g.lower_only && !g.skip_upper;
        if ((h->tune.sk == 2 || (plain && g.K >= 64 * BK && eff < (long)h->tune.sk_rounds * G && eff % G > G / 16 && eff % G < G - G / 4)) && gpk_i_splitk_reserve(h) == 0) {
            SkPlan* p = sk_plan_for<BM, BN>(h, g, G);
            if (p) {
                g.sk_segs = p->d_segs; g.sk_off = p->d_off; g.ws = h->d_splitk_ws; g.cnt = h->d_splitk_cnt;
                if (h->tune.sk_stagger > 0) g.stagger = h->tune.sk_stagger;
                dim3 sgrid(p->nblocks);
                if (g.tri_a) gemm_f64_kernel<BM, BN, WM, WN, false, false, true, true><<<sgrid, block, dyn, h->stream>>>(g);
                else if (!ta && !tb) gemm_f64_kernel<BM, BN, WM, WN, false, false, false, true><<<sgrid, block, dyn, h->stream>>>(g);
                else if (!ta && tb) gemm_f64_kernel<BM, BN, WM, WN, false, true, false, true><<<sgrid, block, dyn, h->stream>>>(g);
                else if (ta && !tb) gemm_f64_kernel<BM, BN, WM, WN, true, false, false, true><<<sgrid, block, dyn, h->stream>>>(g);
                else gemm_f64_kernel<BM, BN, WM, WN, true, true, false, true><<<sgrid, block, dyn, h->stream>>>(g);
                GPK_LAUNCH_CHECK(h);
                return 0;
            }
        }
    }
    if (g.tri_a) gemm_f64_kernel<BM, BN, WM, WN, false, false, true><<<grid, block, dyn, h->stream>>>(g);   // (only NN reaches here)
    else if (!ta && !tb) gemm_f64_kernel<BM, BN, WM, WN, false, false><<<grid, block, dyn, h->stream>>>(g);
    else if (!ta && tb) gemm_f64_kernel<BM, BN, WM, WN, false, true><<<grid, block, dyn, h->stream>>>(g);
    else if (ta && !tb) gemm_f64_kernel<BM, BN, WM, WN, true, false><<<grid, block, dyn, h->stream>>>(g);
    else gemm_f64_kernel<BM, BN, WM, WN, true, true><<<grid, block, dyn, h->stream>>>(g);
    GPK_LAUNCH_CHECK(h);
    return 0;
}

__global__ void symmetrize_kernel(double* A, int n, long lda) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i < n && j < n && j > i) A[(long)i * lda + j] = A[(long)j * lda + i];
}

__global__ void tril_kernel(double* A, int n, long lda) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i < n && j < n && j > i) A[(long)i * lda + j] = 0.0;
}

}  // namespace




void gpk_i_sk_free(gpk_handle h) {
    SkCache* c = (SkCache*)h->sk_cache;
    if (!c) return;
    for (SkPlan* p : c->plans) delete p;
    if (c->arena) (void)hipFree(c->arena);
    delete c;
    h->sk_cache = nullptr;
}

int gpk_i_gemm(gpk_handle h, bool ta, bool tb, int m, int n, int k, double alpha, const double* A, int lda,
               const double* B, int ldb, double beta, double* C, int ldc, bool lower_only, int lead, bool tri_a, bool skip_upper) {
    if (m <= 0 || n <= 0) return 0;
    if (k < 0 || !A || !B || !C) return gpk_bad_arg(h, "gemm: sizes/pointers");
    if (lower_only && m != n) return gpk_bad_arg(h, "gemm: lower_only needs a square C");
    GemmArgs g;
    g.M = m; g.N = n; g.K = k; g.alpha = alpha; g.beta = beta;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.lower_only = lower_only ? 1 : 0;
    g.lead = (lead > 0 && !tb && (!lower_only || (ta && A == B))) ? lead : 0;
    g.lead_div = h->lead_div > 0 ? h->lead_div : 1;
    g.stair = GpkStair(); g.stair_col0 = g.stair_row0 = 0;
    if (g.lead > 0 && h->stair.nseg > 0) { g.stair = h->stair; g.stair_col0 = h->stair_col0; g.stair_row0 = h->stair_row0; }
    g.tri_a = (tri_a && !ta && !tb && !lower_only) ? 1 : 0;
    g.skip_upper = (skip_upper && !lower_only) ? 1 : 0;
    g.stagger = h->tune.stagger;
    g.band = 0; g.splitk = 1; g.ws = nullptr; g.cnt = nullptr; g.nsuper = 0; g.ntm_full = 0;   // (set per launch configuration below)
    g.sk_segs = nullptr; g.sk_off = nullptr; g.row_order = 0;
    g.rev_k = (h->tune.rev_k && g.lead > 0) ? 1 : 0;
    g.vecA = ((lda & 1) == 0) && (((uintptr_t)A & 15) == 0);
    g.vecB = ((ldb & 1) == 0) && (((uintptr_t)B & 15) == 0);
    if (k <= 64 && !lower_only && !g.tri_a && !g.skip_upper && h->tune.force_cfg == 0 &&
        (beta == 0.0 || (beta == 1.0 && (alpha == 1.0 || alpha == -1.0))))
        return launch_k64(h, ta, tb, g);
    // The 64x64 configuration (4 workgroups per CU, two slabs in flight) is used for every shape: with the straight-line
    // two-slab prefetch it beats the 128x128 one (2 per CU, 232 VGPRs, one slab in flight) from 2048 to 21000 on every
    // operand layout (tools/gemm_big_probe.py: 62.0 vs 58.8 TF/s NN 10500^3-ish, 58.6 vs 49.8 NT K=512).  The large
    // tiles stay reachable through gpk_debug_set(0, 1) as the reference point for a register-leaner rewrite.
    const bool big = (h->tune.force_cfg == 1) && !g.tri_a;
    if (big) return launch_cfg<128, 128, 64, 64>(h, ta, tb, g);
    // (lower-triangular output with leading zeros = the product S^T S: from ~8000 lower 64 x 64 tiles on -- n_z ~ 8000 -- the 128 x 128 tile's
    // halved operand traffic wins: north-star size 23.9 -> 22.2 ms; at config 2, 2016 tiles, it loses: 1.59 -> 2.07 ms)
    const bool big_lower = lower_only && g.lead > 0 && h->tune.force_cfg == 0 && h->tune.big_lower_min > 0 && (long)gpk_ceil_div(m, 64) * (gpk_ceil_div(m, 64) + 1) / 2 >= h->tune.big_lower_min;
    if ((h->tune.force_cfg == 4 || big_lower || (h->tune.force_cfg == 0 && h->tune.big_min > 0 && k > 64 && (long)gpk_ceil_div(m, 64) * gpk_ceil_div(n, 64) >= h->tune.big_min)) && !g.tri_a && (!lower_only || h->tune.force_cfg == 4 || big_lower))
        return launch_cfg<128, 128, 32, 32>(h, ta, tb, g);           // 16 waves, one workgroup per CU
    if ((h->tune.force_cfg == 3 || (h->tune.force_cfg == 0 && h->tune.tall_min > 0 && (long)gpk_ceil_div(m, 64) * gpk_ceil_div(n, 64) >= h->tune.tall_min)) && !g.tri_a && !lower_only)
        return launch_cfg<128, 64, 32, 32>(h, ta, tb, g);            // 8 waves, 2 workgroups per CU
    // short-and-wide updates of the triangular-solve recursion (M = 256 or 512 against ~4000 columns): 64x64 tiles give
    // only 1-2 workgroups per CU, i.e. one wave per SIMD and nothing to hide latency behind; 32x64 tiles double that
    long t64 = (long)gpk_ceil_div(m, 64) * gpk_ceil_div(n, 64);
    if (g.tri_a) t64 /= 2;                                            // workgroups handle pairs of row tiles
    // (also for long K: restricting this to K <= 1024 was measured slower on the 512-column products of the pipelined SYRK and on
    // the mid-size updates of the triangular solve -- 504 tiles of 64x64 leave the CUs at 2-3 workgroups)
    if (h->tune.force_cfg == 0 && h->tile_req == 128 && !lower_only && !g.tri_a) return launch_cfg<128, 64, 32, 32>(h, ta, tb, g);
    if (h->tune.force_cfg == 0 && !lower_only && t64 < 2 * h->num_cu && m >= 64 && h->tile_req != 64) return launch_cfg<32, 64, 16, 32>(h, ta, tb, g);
    // (A "round model" -- co-resident workgroups start and finish together, a partly filled last round costs at least half a round, so
    // e.g. 1260 tiles of 64 rows should lose against 2457 tiles of 32 rows -- was tried as the selector and is wrong for this kernel:
    // 383 -> 420 us for that launch, 1384 -> 1477 us for the 3276-tile one; only launches below 0.6 rounds gained, 121 -> 105 us.)
    return launch_cfg<64, 64, 32, 32>(h, ta, tb, g);
}

extern "C" int gpk_gemm(gpk_handle h, int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda,
                        const double* B, int ldb, double beta, double* C, int ldc) {
    if (!h) return GPK_ERR_ARG;
    return gpk_i_gemm(h, ta != 0, tb != 0, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, false);
}

extern "C" int gpk_gemm_lz(gpk_handle h, int ta, int m, int n, int k, double alpha, const double* A, int lda,
                           const double* B, int ldb, double beta, double* C, int ldc, int lead) {
    if (!h) return GPK_ERR_ARG;
    return gpk_i_gemm(h, ta != 0, false, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, false, lead > 0 ? lead : 0);
}

extern "C" int gpk_symmetrize_lower(gpk_handle h, double* A, int n, int lda) {
    if (!h || !A) return GPK_ERR_ARG;
    if (n <= 0) return 0;
    dim3 grid(gpk_ceil_div(n, 64), gpk_ceil_div(n, 4));
    symmetrize_kernel<<<grid, 256, 0, h->stream>>>(A, n, lda);
    GPK_LAUNCH_CHECK(h);
    return 0;
}

extern "C" int gpk_tril(gpk_handle h, double* A, int n, int lda) {
    if (!h || !A) return GPK_ERR_ARG;
    if (n <= 0) return 0;
    dim3 grid(gpk_ceil_div(n, 64), gpk_ceil_div(n, 4));
    tril_kernel<<<grid, 256, 0, h->stream>>>(A, n, lda);
    GPK_LAUNCH_CHECK(h);
    return 0;
}

extern "C" int gpk_syrk(gpk_handle h, int n, int k, double alpha, const double* A, int lda, double beta, double* C,
                        int ldc, int full) {
    if (!h) return GPK_ERR_ARG;
    GPK_TRY(gpk_i_gemm(h, true, false, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, true));
    if (full) return gpk_symmetrize_lower(h, C, n, ldc);
    return 0;
}

// gpk_assemble.hip -- fused derivative-kernel Gram block evaluator (HBM-write bound).
//
// Replaces Gram_matrix_assembly / construct_Theta_test (reference src/Gram_matrice.py:11-289) and the nugget of
// *.Gram_matrix (src/PDEs.py:56-73,250-269,391-409; src/InverseProblems.py:66-99).
//
// Math.  kappa = exp(-(p1 d1^2 + p2 d2^2)/2), d = x - y.  With the 1-D Hermite factors
//   h0 = 1, h1 = p d, h2 = p^2 d^2 - p, h3 = p^3 d^3 - 3 p^2 d, h4 = p^4 d^4 - 6 p^3 d^2 + 3 p^2
// every mixed partial is  d_x^alpha d_y^beta kappa = (-1)^{|alpha|} h_{a1+b1}(p1,d1) h_{a2+b2}(p2,d2) kappa.
// A Theta entry is <row functional, column functional> where a functional is a short list of multi-indices
// (delta, d1, d2, d2^2, Laplacian).  One exp per POINT PAIR feeds every block that pair contributes to
// (up to 16 entries), instead of one autodiff'd exp per entry per block as in the reference.
//
// Mapping to the hardware: a workgroup owns TP row points x 256 column points.  Lane <-> column point, so for a
// fixed (row functional, column functional, row point) the 64 lanes of a wave store 512 contiguous bytes; the row
// point is wave-uniform (scalar loads, no LDS needed: 16 B per point).  No symmetry trick: the ALU cost per pair
// (~1 exp + ~60 flops) is >10x below the HBM write time of its 32..128 output bytes.
#include "gpk_common.h"

namespace {

enum { F_DELTA = 0, F_D1 = 1, F_D2 = 2, F_DD2 = 3, F_LAP = 4 };

// multi-index lists of the five functionals
__host__ __device__ constexpr int f_count(int f) { return f == F_LAP ? 2 : 1; }
__host__ __device__ constexpr int f_a1(int f, int i) { return f == F_D1 ? 1 : (f == F_LAP && i == 0 ? 2 : 0); }
__host__ __device__ constexpr int f_a2(int f, int i) { return f == F_D2 ? 1 : (f == F_DD2 ? 2 : (f == F_LAP && i == 1 ? 2 : 0)); }

template <int FX, int FY>
__host__ __device__ __forceinline__ double pair_coeff(const double (&a)[5], const double (&b)[5]) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < f_count(FX); ++i) {
#pragma unroll
        for (int j = 0; j < f_count(FY); ++j) {
            const int n1 = f_a1(FX, i) + f_a1(FY, j);
            const int n2 = f_a2(FX, i) + f_a2(FY, j);
            const double t = a[n1] * b[n2];
            if ((f_a1(FX, i) + f_a2(FX, i)) & 1) s -= t; else s += t;
        }
    }
    return s;
}

__host__ __device__ __forceinline__ void hermite(double p, double d, double (&h)[5]) {
    const double q = p * d;
    const double q2 = q * q;
    h[0] = 1.0;
    h[1] = q;
    h[2] = q2 - p;
    h[3] = q * (q2 - 3.0 * p);
    h[4] = q2 * (q2 - 6.0 * p) + 3.0 * p * p;
}

// layouts: functional of each Theta block and whether it lives on domain points only (0) or domain+boundary (1)
template <int LAYOUT> struct Lay;
template <> struct Lay<GPK_LAYOUT_ELLIPTIC> { static constexpr int nb = 2; static constexpr int f[4] = {F_LAP, F_DELTA, 0, 0};          static constexpr int db[4] = {0, 1, 0, 0}; };
template <> struct Lay<GPK_LAYOUT_BURGERS>  { static constexpr int nb = 4; static constexpr int f[4] = {F_D1, F_D2, F_DD2, F_DELTA};   static constexpr int db[4] = {0, 0, 0, 1}; };
template <> struct Lay<GPK_LAYOUT_EIKONAL>  { static constexpr int nb = 4; static constexpr int f[4] = {F_D1, F_D2, F_LAP, F_DELTA};   static constexpr int db[4] = {0, 0, 0, 1}; };
template <> struct Lay<GPK_LAYOUT_DARCY_A>  { static constexpr int nb = 3; static constexpr int f[4] = {F_D1, F_D2, F_DELTA, 0};       static constexpr int db[4] = {0, 0, 0, 0}; };

struct AsmArgs {
    const double* px; const double* py;   // SoA points: domain first, then boundary
    int Nd, M;                            // M = number of points visited (Nd+Nb, or Nd for DARCY_A)
    double p1, p2;
    double* out; long ld;
    int off[4]; int size[4];
    double nug[4];
    const double* tx; int Nt;             // test mode: (Nt,2) row-major test points
    const double* coeff;                  // extend mode
};

constexpr int TP = 32;                    // row points per workgroup

__global__ void pack_points_kernel(const double* __restrict__ Xd, int Nd, const double* __restrict__ Xb, int Nb,
                                   double* __restrict__ px, double* __restrict__ py) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Nd) { px[i] = Xd[2 * i]; py[i] = Xd[2 * i + 1]; }
    else if (i < Nd + Nb) { px[i] = Xb[2 * (i - Nd)]; py[i] = Xb[2 * (i - Nd) + 1]; }
}

template <int L, int BI, int BJ>
__device__ __forceinline__ void store_block(const AsmArgs& g, int p, int q, const double (&a)[5], const double (&b)[5], double e) {
    if (q < g.size[BJ]) {
        double v = pair_coeff<Lay<L>::f[BI], Lay<L>::f[BJ]>(a, b) * e;
        if (BI == BJ && p == q) v += g.nug[BI];
        g.out[(long)(g.off[BI] + p) * g.ld + g.off[BJ] + q] = v;
    }
}

template <int L, int BI>
__device__ __forceinline__ void store_row(const AsmArgs& g, int p, int q, const double (&a)[5], const double (&b)[5], double e) {
    if (p < g.size[BI]) {                               // wave-uniform
        store_block<L, BI, 0>(g, p, q, a, b, e);
        if (Lay<L>::nb > 1) store_block<L, BI, 1>(g, p, q, a, b, e);
        if (Lay<L>::nb > 2) store_block<L, BI, 2>(g, p, q, a, b, e);
        if (Lay<L>::nb > 3) store_block<L, BI, 3>(g, p, q, a, b, e);
    }
}

// Two column points per lane, one 16-byte store per (row functional, column functional, row point): a wave writes 1 KB contiguous
// per store instruction and issues half as many of them.  Needs every block offset, the leading dimension and the base address to
// be even multiples of 8 bytes (checked by the launcher; otherwise the one-point-per-lane kernel below runs).
typedef double asm_d2 __attribute__((ext_vector_type(2)));

template <int L, int BI, int BJ, int NT>
__device__ __forceinline__ void store_block2(const AsmArgs& g, int p, int q, const double (&a0)[5], const double (&b0)[5], double e0,
                                             const double (&a1)[5], const double (&b1)[5], double e1) {
    if (q < g.size[BJ]) {                                            // (sizes are even here: q and q + 1 are both inside or both outside)
        double v0 = pair_coeff<Lay<L>::f[BI], Lay<L>::f[BJ]>(a0, b0) * e0;
        double v1 = pair_coeff<Lay<L>::f[BI], Lay<L>::f[BJ]>(a1, b1) * e1;
        if (BI == BJ) { if (p == q) v0 += g.nug[BI]; if (p == q + 1) v1 += g.nug[BI]; }
        asm_d2* dst = reinterpret_cast<asm_d2*>(g.out + (long)(g.off[BI] + p) * g.ld + g.off[BJ] + q);
        // NT (gpk_tune key 55): Theta is written once and not read by this kernel -- a non-temporal store (global_store_dwordx4 ... nt)
        // tells L2 / the Infinity Cache not to keep the line
        // (2 / 3: write-through scopes sc0 sc1 without / with nt, inline asm -- measured next to it, see the table at gpk_assemble below)
        const asm_d2 v = (asm_d2){v0, v1};
        if (NT == 1) __builtin_nontemporal_store(v, dst);
        else if (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(v) : "memory");
        else if (NT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(dst), "v"(v) : "memory");
        else *dst = v;
    }
}

template <int L, int BI, int NT>
__device__ __forceinline__ void store_row2(const AsmArgs& g, int p, int q, const double (&a0)[5], const double (&b0)[5], double e0,
                                           const double (&a1)[5], const double (&b1)[5], double e1) {
    if (p < g.size[BI]) {                               // wave-uniform
        store_block2<L, BI, 0, NT>(g, p, q, a0, b0, e0, a1, b1, e1);
        if (Lay<L>::nb > 1) store_block2<L, BI, 1, NT>(g, p, q, a0, b0, e0, a1, b1, e1);
        if (Lay<L>::nb > 2) store_block2<L, BI, 2, NT>(g, p, q, a0, b0, e0, a1, b1, e1);
        if (Lay<L>::nb > 3) store_block2<L, BI, 3, NT>(g, p, q, a0, b0, e0, a1, b1, e1);
    }
}

template <int L, int NT>
__global__ __launch_bounds__(256) void assemble2_kernel(AsmArgs g) {
    const int q = 2 * (blockIdx.x * 256 + threadIdx.x);
    const bool live = q < g.M;                                       // (M even: q + 1 < M as well)
    const double y1a = live ? g.px[q] : 0.0, y2a = live ? g.py[q] : 0.0;
    const double y1b = live ? g.px[q + 1] : 0.0, y2b = live ? g.py[q + 1] : 0.0;
    const int p0 = blockIdx.y * TP;
    const int pend = min(p0 + TP, g.M);
    for (int p = p0; p < pend; ++p) {
        const double x1 = g.px[p], x2 = g.py[p];        // uniform address -> scalar loads
        if (!live) continue;
        const double d1a = x1 - y1a, d2a = x2 - y2a, d1b = x1 - y1b, d2b = x2 - y2b;
        const double e0 = exp(-0.5 * (g.p1 * d1a * d1a + g.p2 * d2a * d2a));
        const double e1 = exp(-0.5 * (g.p1 * d1b * d1b + g.p2 * d2b * d2b));
        double a0[5], b0[5], a1[5], b1[5];
        hermite(g.p1, d1a, a0); hermite(g.p2, d2a, b0);
        hermite(g.p1, d1b, a1); hermite(g.p2, d2b, b1);
        store_row2<L, 0, NT>(g, p, q, a0, b0, e0, a1, b1, e1);
        if (Lay<L>::nb > 1) store_row2<L, 1, NT>(g, p, q, a0, b0, e0, a1, b1, e1);
        if (Lay<L>::nb > 2) store_row2<L, 2, NT>(g, p, q, a0, b0, e0, a1, b1, e1);
        if (Lay<L>::nb > 3) store_row2<L, 3, NT>(g, p, q, a0, b0, e0, a1, b1, e1);
    }
}

template <int L>
__global__ __launch_bounds__(256) void assemble_kernel(AsmArgs g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool live = q < g.M;
    const double y1 = live ? g.px[q] : 0.0, y2 = live ? g.py[q] : 0.0;
    const int p0 = blockIdx.y * TP;
    const int pend = min(p0 + TP, g.M);
    for (int p = p0; p < pend; ++p) {
        const double x1 = g.px[p], x2 = g.py[p];        // uniform address -> scalar loads
        if (!live) continue;
        const double d1 = x1 - y1, d2 = x2 - y2;
        const double e = exp(-0.5 * (g.p1 * d1 * d1 + g.p2 * d2 * d2));
        double a[5], b[5];
        hermite(g.p1, d1, a);
        hermite(g.p2, d2, b);
        store_row<L, 0>(g, p, q, a, b, e);
        if (Lay<L>::nb > 1) store_row<L, 1>(g, p, q, a, b, e);
        if (Lay<L>::nb > 2) store_row<L, 2>(g, p, q, a, b, e);
        if (Lay<L>::nb > 3) store_row<L, 3>(g, p, q, a, b, e);
    }
}

// test rows: functional delta at the test point, column functionals of the layout
template <int L, int BJ>
__device__ __forceinline__ void store_test(const AsmArgs& g, int t, int q, const double (&a)[5], const double (&b)[5], double e) {
    if (q < g.size[BJ]) g.out[(long)t * g.ld + g.off[BJ] + q] = pair_coeff<F_DELTA, Lay<L>::f[BJ]>(a, b) * e;
}

template <int L>
__global__ __launch_bounds__(256) void assemble_test_kernel(AsmArgs g) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= g.M) return;
    const double y1 = g.px[q], y2 = g.py[q];
    const int t0 = blockIdx.y * TP;
    const int tend = min(t0 + TP, g.Nt);
    for (int t = t0; t < tend; ++t) {
        const double d1 = g.tx[2 * t] - y1, d2 = g.tx[2 * t + 1] - y2;
        const double e = exp(-0.5 * (g.p1 * d1 * d1 + g.p2 * d2 * d2));
        double a[5], b[5];
        hermite(g.p1, d1, a);
        hermite(g.p2, d2, b);
        store_test<L, 0>(g, t, q, a, b, e);
        if (Lay<L>::nb > 1) store_test<L, 1>(g, t, q, a, b, e);
        if (Lay<L>::nb > 2) store_test<L, 2>(g, t, q, a, b, e);
        if (Lay<L>::nb > 3) store_test<L, 3>(g, t, q, a, b, e);
    }
}

template <int L, int BJ>
__device__ __forceinline__ double acc_test(const AsmArgs& g, int q, const double (&a)[5], const double (&b)[5]) {
    return (q < g.size[BJ]) ? pair_coeff<F_DELTA, Lay<L>::f[BJ]>(a, b) * g.coeff[g.off[BJ] + q] : 0.0;
}

// out[t] = sum_c Theta_test[t, c] * coeff[c]; one workgroup per test point, lanes stride over column points.
template <int L>
__global__ __launch_bounds__(256) void extend_kernel(AsmArgs g) {
    __shared__ double red[4];
    const int t = blockIdx.x;
    const double x1 = g.tx[2 * t], x2 = g.tx[2 * t + 1];
    double s = 0.0;
    for (int q = threadIdx.x; q < g.M; q += 256) {
        const double d1 = x1 - g.px[q], d2 = x2 - g.py[q];
        const double e = exp(-0.5 * (g.p1 * d1 * d1 + g.p2 * d2 * d2));
        double a[5], b[5];
        hermite(g.p1, d1, a);
        hermite(g.p2, d2, b);
        double v = acc_test<L, 0>(g, q, a, b);
        if (Lay<L>::nb > 1) v += acc_test<L, 1>(g, q, a, b);
        if (Lay<L>::nb > 2) v += acc_test<L, 2>(g, q, a, b);
        if (Lay<L>::nb > 3) v += acc_test<L, 3>(g, q, a, b);
        s += v * e;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) g.out[t] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int L> void fill_layout(AsmArgs& g, int Nd, int Nb) {
    int o = 0;
    for (int b = 0; b < 4; ++b) {
        g.size[b] = b < Lay<L>::nb ? (Lay<L>::db[b] ? Nd + Nb : Nd) : 0;
        g.off[b] = o;
        o += g.size[b];
    }
}

int fill_common(gpk_handle h, AsmArgs& g, int layout, int kernel, const double* kp, const double* Xd, int Nd,
                const double* Xb, int Nb) {
    if (Nd <= 0 || Nb < 0 || !kp || !Xd) return gpk_bad_arg(h, "assemble: sizes/pointers");
    if (kernel == GPK_KERNEL_GAUSSIAN) { g.p1 = g.p2 = 1.0 / (kp[0] * kp[0]); }                 // src/kernels.py:12-13
    else if (kernel == GPK_KERNEL_ANISOTROPIC) { g.p1 = 2.0 / (kp[0] * kp[0]); g.p2 = 2.0 / (kp[1] * kp[1]); }   // :95-99
    else return gpk_bad_arg(h, "assemble: kernel id");
    switch (layout) {
        case GPK_LAYOUT_ELLIPTIC: fill_layout<GPK_LAYOUT_ELLIPTIC>(g, Nd, Nb); break;
        case GPK_LAYOUT_BURGERS:  fill_layout<GPK_LAYOUT_BURGERS>(g, Nd, Nb); break;
        case GPK_LAYOUT_EIKONAL:  fill_layout<GPK_LAYOUT_EIKONAL>(g, Nd, Nb); break;
        case GPK_LAYOUT_DARCY_A:  fill_layout<GPK_LAYOUT_DARCY_A>(g, Nd, Nb); break;
        default: return gpk_bad_arg(h, "assemble: layout id");
    }
    const int Mall = Nd + Nb;
    GPK_TRY(gpk_i_ensure_points(h, 2 * (size_t)Mall));
    g.px = h->d_pts; g.py = h->d_pts + Mall;
    pack_points_kernel<<<gpk_ceil_div(Mall, 256), 256, 0, h->stream>>>(Xd, Nd, Xb, Nb, h->d_pts, h->d_pts + Mall);
    GPK_LAUNCH_CHECK(h);
    g.Nd = Nd;
    g.M = (layout == GPK_LAYOUT_DARCY_A) ? Nd : Mall;
    for (int b = 0; b < 4; ++b) g.nug[b] = 0.0;
    g.tx = nullptr; g.Nt = 0; g.coeff = nullptr;
    return 0;
}

// value of <f, f> at d = 0 (SURVEY §8a-K last column): makes the adaptive trace ratios analytic
template <int L> void diag_values(double p1, double p2, double (&c)[4]) {
    double a[5], b[5];
    hermite(p1, 0.0, a);
    hermite(p2, 0.0, b);
    c[0] = pair_coeff<Lay<L>::f[0], Lay<L>::f[0]>(a, b);
    c[1] = pair_coeff<Lay<L>::f[1], Lay<L>::f[1]>(a, b);
    c[2] = pair_coeff<Lay<L>::f[2], Lay<L>::f[2]>(a, b);
    c[3] = pair_coeff<Lay<L>::f[3], Lay<L>::f[3]>(a, b);
}

}  // namespace


extern "C" int gpk_assemble(gpk_handle h, int layout, int kernel, const double* kp, const double* Xd, int Nd,
                            const double* Xb, int Nb, double nugget, int nugget_type, double* Theta, int ld,
                            double* ratios) {
    if (!h || !Theta) return GPK_ERR_ARG;
    AsmArgs g;
    GPK_TRY(fill_common(h, g, layout, kernel, kp, Xd, Nd, Xb, Nb));
    const int nb = (layout == GPK_LAYOUT_ELLIPTIC) ? 2 : (layout == GPK_LAYOUT_DARCY_A ? 3 : 4);
    const int N = g.off[nb - 1] + g.size[nb - 1];
    if (ld < N) return gpk_bad_arg(h, "assemble: ld < N");
    double c[4] = {0, 0, 0, 0};
    switch (layout) {
        case GPK_LAYOUT_ELLIPTIC: diag_values<GPK_LAYOUT_ELLIPTIC>(g.p1, g.p2, c); break;
        case GPK_LAYOUT_BURGERS:  diag_values<GPK_LAYOUT_BURGERS>(g.p1, g.p2, c); break;
        case GPK_LAYOUT_EIKONAL:  diag_values<GPK_LAYOUT_EIKONAL>(g.p1, g.p2, c); break;
        case GPK_LAYOUT_DARCY_A:  diag_values<GPK_LAYOUT_DARCY_A>(g.p1, g.p2, c); break;
    }
    // ratio_k = trace(block k) / trace(last block)   (src/PDEs.py:62-66, 256-262, 397-402; IP.py:72-87)
    double r[4] = {0, 0, 0, 1.0};
    const double tr_last = (double)g.size[nb - 1] * c[nb - 1];
    for (int b = 0; b < nb - 1; ++b) r[b] = ((double)g.size[b] * c[b]) / tr_last;
    if (ratios) { ratios[0] = ratios[1] = ratios[2] = 0.0; for (int b = 0; b < nb - 1; ++b) ratios[b] = r[b]; }
    for (int b = 0; b < nb; ++b) {
        if (nugget_type == GPK_NUGGET_ADAPTIVE) g.nug[b] = nugget * (b == nb - 1 ? 1.0 : r[b]);
        else if (nugget_type == GPK_NUGGET_IDENTITY) g.nug[b] = nugget;
        else if (nugget_type == GPK_NUGGET_NONE) g.nug[b] = 0.0;
        else return gpk_bad_arg(h, "assemble: nugget_type");
    }
    g.out = Theta; g.ld = ld;
    // two column points per lane (16-byte stores) when every pair (q, q + 1) stays inside one block and is 16-byte aligned
    bool pairs = h->tune.asm_pairs && (ld % 2 == 0) && (((uintptr_t)Theta & 15) == 0) && (g.M % 2 == 0);
    for (int b = 0; b < nb; ++b) pairs = pairs && (g.off[b] % 2 == 0) && (g.size[b] % 2 == 0);
    // (per-phase timing on: HIP events around the evaluator launch alone -- the point packing and the host work above stay outside)
    if (h->prof) {
        if (!h->asm_ev[0]) for (int i = 0; i < 2; ++i) GPK_HIP(h, hipEventCreate(&h->asm_ev[i]));
        GPK_HIP(h, hipEventRecord(h->asm_ev[0], h->stream));
    }
    struct AsmStop {
        gpk_handle h; ~AsmStop() { if (h->prof && h->asm_ev[1]) h->asm_timed = hipEventRecord(h->asm_ev[1], h->stream) == hipSuccess; }
    } asm_stop{h};
    if (pairs) {
        dim3 grid2(gpk_ceil_div(g.M / 2, 256), gpk_ceil_div(g.M, TP));
        auto launch = [&](auto nt_c) {
            constexpr int NT = decltype(nt_c)::value;
            switch (layout) {
                case GPK_LAYOUT_ELLIPTIC: assemble2_kernel<GPK_LAYOUT_ELLIPTIC, NT><<<grid2, 256, 0, h->stream>>>(g); break;
                case GPK_LAYOUT_BURGERS:  assemble2_kernel<GPK_LAYOUT_BURGERS, NT><<<grid2, 256, 0, h->stream>>>(g); break;
                case GPK_LAYOUT_EIKONAL:  assemble2_kernel<GPK_LAYOUT_EIKONAL, NT><<<grid2, 256, 0, h->stream>>>(g); break;
                case GPK_LAYOUT_DARCY_A:  assemble2_kernel<GPK_LAYOUT_DARCY_A, NT><<<grid2, 256, 0, h->stream>>>(g); break;
            }
        };
        switch (h->tune.asm_nt) {
            case 1: launch(std::integral_constant<int, 1>{}); break;
            case 2: launch(std::integral_constant<int, 2>{}); break;
            case 3: launch(std::integral_constant<int, 3>{}); break;
            default: launch(std::integral_constant<int, 0>{}); break;
        }
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    dim3 grid(gpk_ceil_div(g.M, 256), gpk_ceil_div(g.M, TP));
    switch (layout) {
        case GPK_LAYOUT_ELLIPTIC: assemble_kernel<GPK_LAYOUT_ELLIPTIC><<<grid, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_BURGERS:  assemble_kernel<GPK_LAYOUT_BURGERS><<<grid, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_EIKONAL:  assemble_kernel<GPK_LAYOUT_EIKONAL><<<grid, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_DARCY_A:  assemble_kernel<GPK_LAYOUT_DARCY_A><<<grid, 256, 0, h->stream>>>(g); break;
    }
    GPK_LAUNCH_CHECK(h);
    return 0;
}

extern "C" int gpk_assemble_test(gpk_handle h, int layout, int kernel, const double* kp, const double* Xt, int Nt,
                                 const double* Xd, int Nd, const double* Xb, int Nb, double* out, int ld) {
    if (!h || !out || !Xt || Nt <= 0) return GPK_ERR_ARG;
    AsmArgs g;
    GPK_TRY(fill_common(h, g, layout, kernel, kp, Xd, Nd, Xb, Nb));
    g.out = out; g.ld = ld; g.tx = Xt; g.Nt = Nt;
    dim3 grid(gpk_ceil_div(g.M, 256), gpk_ceil_div(Nt, TP));
    switch (layout) {
        case GPK_LAYOUT_ELLIPTIC: assemble_test_kernel<GPK_LAYOUT_ELLIPTIC><<<grid, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_BURGERS:  assemble_test_kernel<GPK_LAYOUT_BURGERS><<<grid, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_EIKONAL:  assemble_test_kernel<GPK_LAYOUT_EIKONAL><<<grid, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_DARCY_A:  assemble_test_kernel<GPK_LAYOUT_DARCY_A><<<grid, 256, 0, h->stream>>>(g); break;
    }
    GPK_LAUNCH_CHECK(h);
    return 0;
}

// err[i] = |truth[i] - approx[i]|, its maximum and its sum of squares in one pass (one workgroup: n is a point count, at most a
// few 10^4; fixed summation order, so the figures are reproducible)
__global__ __launch_bounds__(1024) void error_metrics_kernel(int n, const double* __restrict__ truth, const double* __restrict__ approx,
                                                              double* __restrict__ err, double* __restrict__ out2) {
    __shared__ double smax[16], ssum[16];
    double m = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double e = fabs(truth[i] - approx[i]);
        if (err) err[i] = e;
        m = (e > m || e != e) ? e : m;                               // NaNs propagate, as under jnp.max
        q = fma(e, e, q);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double mo = __shfl_down(m, off, 64);
        m = (mo > m || mo != mo) ? mo : m;
        q += __shfl_down(q, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = m; ssum[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double mm = 0.0, qq = 0.0;
        for (int w = 0; w < 16; ++w) { mm = (smax[w] > mm || smax[w] != smax[w]) ? smax[w] : mm; qq += ssum[w]; }
        out2[0] = mm; out2[1] = qq;
    }
}

extern "C" int gpk_error_metrics(gpk_handle h, int n, const double* truth, const double* approx, double* err_all,
                                 double* host_max, double* host_l2) {
    if (!h || !truth || !approx || n <= 0 || !host_max || !host_l2) return GPK_ERR_ARG;
    error_metrics_kernel<<<1, 1024, 0, h->stream>>>(n, truth, approx, err_all, h->d_scalars + 2);
    GPK_LAUNCH_CHECK(h);
    double r[2] = {0.0, 0.0};
    GPK_HIP(h, hipMemcpyAsync(r, h->d_scalars + 2, sizeof r, hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    *host_max = r[0];
    *host_l2 = sqrt(r[1] / (double)n);
    return 0;
}

extern "C" int gpk_extend(gpk_handle h, int layout, int kernel, const double* kp, const double* Xt, int Nt,
                          const double* Xd, int Nd, const double* Xb, int Nb, const double* coeff, double* out) {
    if (!h || !out || !Xt || !coeff || Nt <= 0) return GPK_ERR_ARG;
    AsmArgs g;
    GPK_TRY(fill_common(h, g, layout, kernel, kp, Xd, Nd, Xb, Nb));
    g.out = out; g.ld = 0; g.tx = Xt; g.Nt = Nt; g.coeff = coeff;
    switch (layout) {
        case GPK_LAYOUT_ELLIPTIC: extend_kernel<GPK_LAYOUT_ELLIPTIC><<<Nt, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_BURGERS:  extend_kernel<GPK_LAYOUT_BURGERS><<<Nt, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_EIKONAL:  extend_kernel<GPK_LAYOUT_EIKONAL><<<Nt, 256, 0, h->stream>>>(g); break;
        case GPK_LAYOUT_DARCY_A:  extend_kernel<GPK_LAYOUT_DARCY_A><<<Nt, 256, 0, h->stream>>>(g); break;
    }
    GPK_LAUNCH_CHECK(h);
    return 0;
}

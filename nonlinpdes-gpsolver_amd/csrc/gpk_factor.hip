// gpk_factor.hip -- Cholesky factorisation and triangular solves (fp64), recursive blocking on top of the MFMA GEMM.
//
// Replaces jnp.linalg.cholesky (reference src/PDEs.py:77,273,413; src/InverseProblems.py:102-103) and every
// jnp.linalg.solve with the triangular factor (src/PDEs.py:86,97,143,161,205,288,306,347,429,450,502;
// src/InverseProblems.py:118-119,145-146,190,195).  The reference calls a GENERAL LU solve on the lower-triangular
// L each time; here L is used as what it is.
//
// Structure: all O(n^3) work is delegated to gpk_i_gemm (MFMA); only 64-wide diagonal blocks are handled by the
// substitution kernels below (true substitution, no explicit inverses: cond(Theta) ~ 1e13+ leaves no slack).
//   potrf(A)      = potrf(A11); A21 <- A21 L11^{-T}; A22 -= A21 A21^T (lower tiles only); potrf(A22)
//   trsm_left(L)  = solve with L11; B2 -= L21 X1; solve with L22          (transposed: mirror image)
//   trsm_right_lt = X1 <- X1 L11^{-T}; X2 -= X1 L21^T; X2 <- X2 L22^{-T}
// The recursion splits at multiples of 64/128 so that sub-blocks stay 16-byte aligned for the GEMM's vector loads.
#include "gpk_common.h"

namespace {

constexpr int NB = 64;

__device__ __forceinline__ double bcast(double v, int src_lane) { return __shfl(v, src_lane, 64); }

// ---- 64x64 (or smaller) Cholesky by ONE wave: lane i owns row i in registers ---------------------------------
__global__ __launch_bounds__(64) void potf2_kernel(double* __restrict__ A, long lda, int n, int* info, int pivot_base) {
    __shared__ double tile[NB * (NB + 1)];
    __shared__ double col[NB];
    const int lane = threadIdx.x;
    for (int r = 0; r < NB; ++r) {                                   // coalesced rows -> LDS (identity padding)
        double v = (r == lane) ? 1.0 : 0.0;
        if (r < n && lane < n) v = A[(long)r * lda + lane];
        tile[r * (NB + 1) + lane] = v;
    }
    __syncthreads();
    double a[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) a[j] = tile[lane * (NB + 1) + j];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double d = bcast(a[j], j);                             // pivot a_jj lives in lane j
        if (!(d > 0.0) && bad == 0) bad = j + 1;                     // NaN-safe test; wave-uniform
        const double s = sqrt(d);
        const double lij = (lane == j) ? s : a[j] / s;
        a[j] = lij;
        col[lane] = lij;
        __syncthreads();
#pragma unroll
        for (int k = j + 1; k < NB; ++k) a[k] -= lij * col[k];       // rank-1 update of row `lane`
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) tile[lane * (NB + 1) + j] = a[j];
    __syncthreads();
    for (int r = 0; r < n; ++r)
        if (lane <= r) A[(long)r * lda + lane] = tile[r * (NB + 1) + lane];
    if (bad && bad <= n && lane == 0) atomicCAS(info, 0, pivot_base + bad);
}

// ---- substitution with a <=64-wide diagonal block, one RHS column (or row) per lane ---------------------------
// TRANS=false: L X = B;  TRANS=true: L^T X = B.
// ROWVEC=false: element (j, c) of B at B[j*ldb + c]  (left solve: lanes = consecutive columns, coalesced)
// ROWVEC=true : element (j, c) of B at B[c*ldb + j]  (right solve X L^T = A viewed as L X^T = A^T; rows of X are
//               staged through LDS so that global traffic stays coalesced)
template <bool TRANS, bool ROWVEC>
__global__ __launch_bounds__(64) void trsm_base_kernel(const double* __restrict__ L, long ldl, int nb,
                                                       double* __restrict__ B, long ldb, int ncols) {
    constexpr int WS = NB + 2;
    __shared__ __attribute__((aligned(16))) double W[NB * WS];      // W[i][j] = coefficient of x_i in equation j
    __shared__ double dg[NB];
    __shared__ double T[ROWVEC ? NB * (NB + 1) : 1];
    const int lane = threadIdx.x;
    const int c0 = blockIdx.x * NB;
    for (int r = 0; r < NB; ++r) {
        double v = 0.0;
        if (r < nb && lane < nb) v = L[(long)r * ldl + lane];
        if (r == lane) dg[r] = (r < nb) ? v : 1.0;
        if (TRANS) W[r * WS + lane] = (lane < r) ? v : 0.0;          // x_r enters equation `lane` (< r) with L[r][lane]
        else       W[lane * WS + r] = (lane < r) ? v : 0.0;          // x_lane enters equation r (> lane) with L[r][lane]
    }
    double x[NB];
    const int c = c0 + lane;
    if (ROWVEC) {
        for (int r = 0; r < NB; ++r) {
            double v = 0.0;
            if (c0 + r < ncols && lane < nb) v = B[(long)(c0 + r) * ldb + lane];
            T[r * (NB + 1) + lane] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB; ++j) x[j] = T[lane * (NB + 1) + j];
    } else {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB; ++j) {                               // clamped address + select: no divergent branches
            const double v = B[(long)min(j, nb - 1) * ldb + min(c, ncols - 1)];
            x[j] = (j < nb) ? v : 0.0;
        }
    }
    const double* Wv = W;
    if (!TRANS) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const double xi = x[i] / dg[i];
            x[i] = xi;
#pragma unroll
            for (int j = i + 1; j < NB; ++j) x[j] -= Wv[i * WS + j] * xi;
            __builtin_amdgcn_sched_barrier(0);                       // keep the LDS reads of step i+1 behind step i
        }
    } else {
#pragma unroll
        for (int i = NB - 1; i >= 0; --i) {
            const double xi = x[i] / dg[i];
            x[i] = xi;
#pragma unroll
            for (int j = 0; j < i; ++j) x[j] -= Wv[i * WS + j] * xi;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (ROWVEC) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB; ++j) T[lane * (NB + 1) + j] = x[j];
        __syncthreads();
        for (int r = 0; r < NB; ++r)
            if (c0 + r < ncols && lane < nb) B[(long)(c0 + r) * ldb + lane] = T[r * (NB + 1) + lane];
    } else {
        // Lanes past the last column redo column ncols-1 and store identical bits (benign): NO conditional block around
        // the stores -- with one, LLVM sinks the whole FMA chain into it, behind every coefficient load, and spills.
        double* out = B + min(c, ncols - 1);
        if (nb == NB) {
#pragma unroll
            for (int j = 0; j < NB; ++j) out[(long)j * ldb] = x[j];
        } else {
            for (int j = 0; j < nb; ++j) {
                double v = x[0];
#pragma unroll
                for (int k = 1; k < NB; ++k) v = (k == j) ? x[k] : v;        // static register indexing
                out[(long)j * ldb] = v;
            }
        }
    }
}

// ---- single-vector triangular solve: 64-wide diagonal block by one wave (lane = equation) ---------------------
template <bool TRANS>
__global__ __launch_bounds__(64) void trsv_diag_kernel(const double* __restrict__ L, long ldl, int nb, double* __restrict__ x) {
    __shared__ double T[NB * (NB + 1)];
    const int lane = threadIdx.x;
    for (int r = 0; r < NB; ++r) {
        double v = (r == lane) ? 1.0 : 0.0;
        if (r < nb && lane < nb) v = L[(long)r * ldl + lane];
        T[r * (NB + 1) + lane] = v;
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
    double s = 0.0;
    for (int j = 0; j < rows; ++j) s += A[(long)j * lda + c] * x[j];
    y[c] -= s;
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

inline int split(int n) {
    // first part: about half, a multiple of 128 when there is room (keeps GEMM operands aligned and tiles full)
    const int q = (n > 256) ? 128 : NB;
    int n1 = ((n / 2 + q - 1) / q) * q;
    if (n1 >= n) n1 = ((n / 2 + NB - 1) / NB) * NB;
    if (n1 >= n) n1 = NB;
    return n1;
}

}  // namespace

int gpk_i_trsm_left(gpk_handle h, bool trans, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
    if (n <= 0 || nrhs <= 0) return 0;
    if (n <= NB) {
        dim3 grid(gpk_ceil_div(nrhs, NB));
        if (trans) trsm_base_kernel<true, false><<<grid, 64, 0, h->stream>>>(L, ldl, n, B, ldb, nrhs);
        else       trsm_base_kernel<false, false><<<grid, 64, 0, h->stream>>>(L, ldl, n, B, ldb, nrhs);
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    const int n1 = split(n), n2 = n - n1;
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

int gpk_i_trsm_right_lt(gpk_handle h, const double* L, int n, int ldl, double* X, int m, int ldx) {
    if (n <= 0 || m <= 0) return 0;
    if (n <= NB) {
        trsm_base_kernel<false, true><<<gpk_ceil_div(m, NB), 64, 0, h->stream>>>(L, ldl, n, X, ldx, m);
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

int gpk_i_potrf(gpk_handle h, double* A, int n, int lda, int pivot_base) {
    if (n <= 0) return 0;
    if (n <= NB) {
        potf2_kernel<<<1, 64, 0, h->stream>>>(A, lda, n, h->d_info, pivot_base);
        GPK_LAUNCH_CHECK(h);
        return 0;
    }
    const int n1 = split(n), n2 = n - n1;
    double* A21 = A + (long)n1 * lda;
    double* A22 = A21 + n1;
    GPK_TRY(gpk_i_potrf(h, A, n1, lda, pivot_base));
    GPK_TRY(gpk_i_trsm_right_lt(h, A, n1, lda, A21, n2, lda));
    GPK_TRY(gpk_i_gemm(h, false, true, n2, n2, n1, -1.0, A21, lda, A21, lda, 1.0, A22, lda, true));
    GPK_TRY(gpk_i_potrf(h, A22, n2, lda, pivot_base + n1));
    return 0;
}

int gpk_i_trsv(gpk_handle h, bool trans, const double* L, int n, int ldl, double* x) {
    if (n <= 0) return 0;
    const int nblk = gpk_ceil_div(n, NB);
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

extern "C" int gpk_trsm(gpk_handle h, int trans, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
    if (!h || !L || !B || n < 0 || nrhs < 0 || ldl < n || ldb < nrhs) return GPK_ERR_ARG;
    if (nrhs == 1 && ldb == 1) return gpk_i_trsv(h, trans != 0, L, n, ldl, B);
    return gpk_i_trsm_left(h, trans != 0, L, n, ldl, B, nrhs, ldb);
}

extern "C" int gpk_potrs(gpk_handle h, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
    GPK_TRY(gpk_trsm(h, 0, L, n, ldl, B, nrhs, ldb));
    return gpk_trsm(h, 1, L, n, ldl, B, nrhs, ldb);
}

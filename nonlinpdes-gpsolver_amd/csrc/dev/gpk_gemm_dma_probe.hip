// gpk_gemm_dma_probe.hip -- EXPERIMENT (round 3, verdict item 3): the operand feed of the fp64 GEMM through LDS-DMA
// (global_load_lds_dwordx4: global memory -> LDS without passing through VGPRs, no ds_write instructions).
//
// Not part of the product path: one plain NN product C = A B (M, N multiples of 64, K a multiple of 16, 16-byte aligned operands)
// with the product kernel's tile (64 x 64, 4 waves of 32 x 32, 16-deep slabs), so that the two feeds can be compared like for like
// (tools/gemm_dma_probe.py).  What changes:
//   * a slab (8 KB of A + 8 KB of B) is written into LDS by 4 + 4 wave-wide DMA instructions per workgroup (two of each per wave);
//     the LDS image of such an instruction is LINEAR (wave-uniform base + lane x 16 bytes), so the bank-conflict-free layout is
//     obtained by permuting the SOURCE addresses: A (k-contiguous rows of 128 bytes): 16-byte granule kq of row r is stored at
//     position kq ^ ((r >> 1) & 7); B (n-contiguous rows of 512 bytes): granule nq of k-row k at nq ^ ((k & 1) << 3).  Every
//     ds_read_b64 of a fragment then touches 32 distinct bank pairs per half-wave.
//   * THREE slab buffers (48 KB), two slabs in flight, ONE raw s_barrier per slab with a counted s_waitcnt vmcnt (the DMA writes
//     are ordered for a reader only by the issuing wave's vmcnt followed by a barrier the reader has passed).
#include "../gpk_common.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ __launch_bounds__(256, 2) void gemm_dma_probe_kernel(int M, int N, int K, const double* __restrict__ A, long lda,
                                                                 const double* __restrict__ B, long ldb, double* __restrict__ C, long ldc) {
    __shared__ __attribute__((aligned(1024))) char smem[3 * 16384];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int li = l & 15, lk = l >> 4;
    const int ntn = N / 64;
    // XCD-aware chunking as in the product kernel's plain order (block b on XCD b % 8)
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int ntm = M / 64;
    const int gid = logical / (8 * ntn), first = gid * 8, gsz = min(ntm - first, 8), rem = logical - gid * 8 * ntn;
    const int tm = first + rem % gsz, tn = rem / gsz;
    const int m0 = tm * 64, n0 = tn * 64;
    const int wm0 = (w >> 1) * 32, wn0 = (w & 1) * 32;

    // ---- DMA sources of my four instructions per slab (permuted granules), advanced by one slab per issue
    const double* srcA[2]; const double* srcB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ia = 2 * w + u;                                    // instruction ia covers rows 8 ia .. 8 ia + 7 of the A tile
        const int row = 8 * ia + (l >> 3), kqp = l & 7;
        srcA[u] = A + (long)(m0 + row) * lda + 2 * (kqp ^ ((row >> 1) & 7));
        const int kr = 2 * ia + (l >> 5), nqp = l & 31;              // instruction ia covers k-rows 2 ia, 2 ia + 1 of the B slab
        srcB[u] = B + (long)kr * ldb + n0 + 2 * (nqp ^ ((kr & 1) << 3));
    }
    auto issue = [&](int kt, int buf) {
        char* base = smem + buf * 16384;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            __builtin_amdgcn_global_load_lds((glb_void*)(srcA[u] + (long)kt * 16), (lds_void*)(base + (2 * w + u) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(srcB[u] + (long)kt * 16 * ldb), (lds_void*)(base + 8192 + (2 * w + u) * 1024), 16, 0, 0);
        }
    };
    // ---- fragment read offsets (bytes inside a slab buffer)
    int offA[2][4], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm0 + 16 * i + li;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) offA[i][ks] = row * 128 + (((2 * ks + (lk >> 1)) ^ ((row >> 1) & 7)) << 4) + (lk & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wn0 + 16 * j + li;
        offB[j] = 8192 + lk * 512 + ((((n >> 1)) ^ ((lk & 1) << 3)) << 4) + (n & 1) * 8;
    }
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    const int nk = K / 16;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // my pieces of slab kt have landed (slab kt+1 may still fly)
        else             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // everybody's pieces of slab kt; everybody done with slab kt-1
        asm volatile("" ::: "memory");
        if (kt + 2 < nk) issue(kt + 2, buf >= 1 ? buf - 1 : 2);       // into the buffer slab kt-1 occupied
        const char* bs = smem + buf * 16384;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const double*>(bs + offA[i][ks]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const double*>(bs + offB[j] + ks * 2048);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            double* crow = C + (long)(m0 + wm0 + 16 * i + lk + 4 * rr) * ldc + n0 + wn0 + li;
#pragma unroll
            for (int j = 0; j < 2; ++j) crow[16 * j] = acc[i][j][rr];
        }
}

}  // namespace

extern "C" int gpk_debug_gemm_dma(gpk_handle h, int m, int n, int k, const double* A, int lda, const double* B, int ldb, double* C, int ldc) {
    if (!h || !A || !B || !C) return GPK_ERR_ARG;
    if (m <= 0 || n <= 0 || k <= 0 || m % 64 || n % 64 || k % 16 || (lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15))
        return gpk_bad_arg(h, "gemm_dma probe: M, N multiples of 64, K of 16, 16-byte aligned operands");
    gemm_dma_probe_kernel<<<(m / 64) * (n / 64), 256, 0, h->stream>>>(m, n, k, A, lda, B, ldb, C, ldc);
    GPK_LAUNCH_CHECK(h);
    return 0;
}

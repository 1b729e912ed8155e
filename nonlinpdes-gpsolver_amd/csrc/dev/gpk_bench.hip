// gpk_bench.hip -- micro-benchmarks that fix the roofline denominators on the box the bench runs on.
//   gpk_ubench_mfma_f64 : issue rate of v_mfma_f64_16x16x4_f64 (the local guides state no fp64 peak)
//   gpk_ubench_hbm_write: streaming 8-byte-per-lane store bandwidth (the assembly kernel's access pattern)
#include "../gpk_common.h"

namespace {
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_f64_kernel(int iters, double* sink) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    // inline asm keeps the four accumulators in place (the builtin form made hipcc shuttle them between VGPRs and
    // AGPRs every iteration, which measured 45 TFLOP/s instead of the issue rate)
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n\t"
                     "v_mfma_f64_16x16x4_f64 %1, %4, %5, %1\n\t"
                     "v_mfma_f64_16x16x4_f64 %2, %4, %5, %2\n\t"
                     "v_mfma_f64_16x16x4_f64 %3, %4, %5, %3\n\t"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const d4 s = c0 + c1 + c2 + c3;
    if (s[0] + s[1] + s[2] + s[3] == -1.0) sink[0] = s[0];
}

__global__ __launch_bounds__(256) void hbm_write_kernel(double* __restrict__ p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = v;
}

// single-wave latency/issue probes (cycles per operation by s_memtime): the diagonal-block kernels are bound by these
__global__ __launch_bounds__(64) void lat_probe_kernel(int mode, double* out, const double* gbuf, int stride) {
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (double)((i + 1) & 1023);
    __syncthreads();
    double x0 = 1.0 + threadIdx.x * 1e-9, x1 = x0, x2 = x0, x3 = x0, x4 = x0, x5 = x0, x6 = x0, x7 = x0;
    const double m = 1.0 - 1e-12, c = 1e-13;
    constexpr int N = 512;
    long t0 = 0, t1 = 0;
    if (mode == 0) {                                                 // dependent v_fma_f64 chain
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(m), "v"(c));
        t1 = clock64();
    } else if (mode == 1) {                                          // 8 independent chains: issue rate
        t0 = clock64();
#pragma unroll 4
        for (int i = 0; i < N / 8; ++i)
            asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                         "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(m), "v"(c));
        t1 = clock64();
    } else if (mode == 2) {                                          // dependent LDS reads (pointer chase): latency
        int idx = threadIdx.x;
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; ++i) idx = (int)lds[idx & 1023];
        t1 = clock64();
        x0 = idx;
    } else if (mode == 3) {                                          // independent broadcast ds_read_b64: issue rate
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; ++i) { x0 += lds[i]; }
        t1 = clock64();
    } else if (mode == 4) {                                          // dependent MFMA chain
        typedef double d4 __attribute__((ext_vector_type(4)));
        d4 acc = {0, 0, 0, 0};
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(m), "v"(c));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        t1 = clock64();
        x0 = acc[0];
    }
    else if (mode == 5 || mode == 6) {                              // dependent global loads: 5 = one line per wave, 6 = 16 lines
        long idx = (mode == 5) ? 0 : (long)(threadIdx.x & 15) * stride + (threadIdx.x >> 4);
        t0 = clock64();
#pragma unroll 8
        for (int i = 0; i < N; ++i) { const double v = gbuf[idx]; idx += (long)v; }   // buffer holds zeros
        t1 = clock64();
        x0 = (double)idx;
    }
    const double s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) { out[0] = (double)(t1 - t0) / N; out[1] = s; }
}

// which XCD does workgroup b land on?  (HW_REG_XCC_ID, bits 3:0)  mode 1: odd workgroups spin ~20 us, even ones exit at once
__global__ __launch_bounds__(256) void xcc_probe_kernel(int* out, int mode) {
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
    if (threadIdx.x == 0) out[blockIdx.x] = xcc;
    if (mode == 1 && (blockIdx.x & 1)) {
        const long t0 = clock64();
        while (clock64() - t0 < 40000) {}
    }
}
}  // namespace

extern "C" int gpk_ubench_mfma_f64(gpk_handle h, int iters, double* host_tflops) {
    if (!h || !host_tflops || iters <= 0) return GPK_ERR_ARG;
    const int blocks = h->num_cu * 4;                        // 4 workgroups x 4 waves per CU = 4 waves per SIMD
    mfma_f64_kernel<<<blocks, 256, 0, h->stream>>>(16, h->d_scalars + 8);      // warm-up
    GPK_TRY(gpk_timer_start(h));
    mfma_f64_kernel<<<blocks, 256, 0, h->stream>>>(iters, h->d_scalars + 8);
    double ms = 0.0;
    GPK_TRY(gpk_timer_stop(h, &ms));
    GPK_LAUNCH_CHECK(h);
    const double flops = (double)blocks * 4.0 * (double)iters * 4.0 * 2.0 * 16 * 16 * 4;
    *host_tflops = flops / (ms * 1e-3) / 1e12;
    return 0;
}

extern "C" int gpk_ubench_hbm_write(gpk_handle h, size_t bytes, int iters, double* host_gbps) {
    if (!h || !host_gbps || bytes < 4096 || iters <= 0) return GPK_ERR_ARG;
    double* buf = nullptr;
    GPK_HIP(h, hipMalloc((void**)&buf, bytes));
    const size_t n = bytes / sizeof(double);
    const int blocks = h->num_cu * 8;
    hbm_write_kernel<<<blocks, 256, 0, h->stream>>>(buf, n, 1.0);
    GPK_TRY(gpk_timer_start(h));
    for (int i = 0; i < iters; ++i) hbm_write_kernel<<<blocks, 256, 0, h->stream>>>(buf, n, (double)i);
    double ms = 0.0;
    GPK_TRY(gpk_timer_stop(h, &ms));
    GPK_HIP(h, hipFree(buf));
    *host_gbps = (double)n * sizeof(double) * iters / (ms * 1e-3) / 1e9;
    return 0;
}

extern "C" int gpk_ubench_latency(gpk_handle h, int mode, double* host_cycles_per_op) {
    if (!h || !host_cycles_per_op) return GPK_ERR_ARG;
    double* gbuf = nullptr;
    GPK_HIP(h, hipMalloc((void**)&gbuf, 16 * 1056 * sizeof(double)));
    GPK_HIP(h, hipMemsetAsync(gbuf, 0, 16 * 1056 * sizeof(double), h->stream));
    lat_probe_kernel<<<1, 64, 0, h->stream>>>(mode, h->d_scalars + 8, gbuf, 1056);
    lat_probe_kernel<<<1, 64, 0, h->stream>>>(mode, h->d_scalars + 8, gbuf, 1056);
    GPK_LAUNCH_CHECK(h);
    GPK_HIP(h, hipMemcpyAsync(host_cycles_per_op, h->d_scalars + 8, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    if (gbuf) GPK_HIP(h, hipFree(gbuf));
    return 0;
}

// census of the CUs a CU-masked stream dispatches to: every workgroup lingers ~20 us (so that the grid spreads over all enabled
// CUs) and reports XCC_ID | HW_ID << 8 (HW_REG_HW_ID: CU_ID bits 11:8, SH_ID 12, SE_ID 15:13 on gfx9)
__global__ void cu_census_kernel(int* out) {
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
    const int hwid = __builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xffff;
    if (threadIdx.x == 0) out[blockIdx.x] = xcc | (hwid << 8);
    const long t0 = clock64();
    while (clock64() - t0 < 40000) {}
}

extern "C" int gpk_ubench_cu_census(gpk_handle h, int first_bit, int nbits, int nblocks, int* host_out) {
    if (!h || !host_out || nblocks <= 0 || first_bit < 0 || nbits <= 0 || first_bit + nbits > 256) return GPK_ERR_ARG;
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = first_bit; i < first_bit + nbits; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    GPK_HIP(h, hipExtStreamCreateWithCUMask(&s, 8, mask));
    int* d = nullptr;
    GPK_HIP(h, hipMalloc((void**)&d, nblocks * sizeof(int)));
    cu_census_kernel<<<nblocks, 256, 0, s>>>(d);
    GPK_LAUNCH_CHECK(h);
    GPK_HIP(h, hipMemcpyAsync(host_out, d, nblocks * sizeof(int), hipMemcpyDeviceToHost, s));
    GPK_HIP(h, hipStreamSynchronize(s));
    GPK_HIP(h, hipFree(d));
    GPK_HIP(h, hipStreamDestroy(s));
    return 0;
}

extern "C" int gpk_ubench_xcc_map(gpk_handle h, int nblocks, int mode, int* host_out) {
    if (!h || !host_out || nblocks <= 0) return GPK_ERR_ARG;
    int* d = nullptr;
    GPK_HIP(h, hipMalloc((void**)&d, nblocks * sizeof(int)));
    xcc_probe_kernel<<<nblocks, 256, 0, h->stream>>>(d, mode);
    GPK_LAUNCH_CHECK(h);
    GPK_HIP(h, hipMemcpyAsync(host_out, d, nblocks * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    GPK_HIP(h, hipFree(d));
    return 0;
}

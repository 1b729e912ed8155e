// gpk_bench.hip -- micro-benchmarks that fix the roofline denominators on the box the bench runs on.
//   gpk_ubench_mfma_f64 : issue rate of v_mfma_f64_16x16x4_f64 (the local guides state no fp64 peak)
//   gpk_ubench_hbm_write: streaming 8-byte-per-lane store bandwidth (the assembly kernel's access pattern)
#include "gpk_common.h"

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

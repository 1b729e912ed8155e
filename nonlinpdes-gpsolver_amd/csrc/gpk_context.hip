// gpk_context.hip -- handle, memory, stream and stopwatch entry points of the C ABI.
#include "gpk_common.h"

int gpk_fail(gpk_handle h, hipError_t e, const char* what, const char* file, int line) {
    if (h) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
        h->err = buf;
    }
    return -(int)e;
}

int gpk_bad_arg(gpk_handle h, const char* what) {
    if (h) h->err = std::string("invalid argument: ") + what;
    return GPK_ERR_ARG;
}

extern "C" const char* gpk_version(void) { return "gpk 0.1 (gfx950, fp64 MFMA)"; }

extern "C" int gpk_create(int device, gpk_handle* out) {
    if (!out) return GPK_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GPK_ERR_NODEV;   // no CPU fallback
    if (device < 0 || device >= count) return GPK_ERR_ARG;
    gpk_ctx* h = new gpk_ctx();
    h->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_info, sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_scalars, 64 * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_pinned, 8 * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemset(h->d_info, 0, sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_obflags, 64 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->d_obflags, 0, 64 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_flags, (GPK_MAX_TRSV_BLOCKS + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->d_flags, 0, (GPK_MAX_TRSV_BLOCKS + 1) * sizeof(int));
    if (e != hipSuccess) { delete h; return -(int)e; }
    h->stream = h->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) h->num_cu = prop.multiProcessorCount;
    // the CU-masked streams of the SYRK/Cholesky pipeline, created NOW (see pipe_setup in gpk_factor.hip: nothing may come between
    // the handle's stream and them); a runtime that refuses CU masks leaves the handle on the one-stream schedule
    if (h->num_cu >= 64 && gpk_i_pipe_streams(h) != 0) {
        h->pipe_unavailable = true;
        h->err.clear();
        (void)hipGetLastError();
    }
    *out = h;
    return 0;
}

extern "C" int gpk_destroy(gpk_handle h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->d_info) (void)hipFree(h->d_info);
    if (h->d_scalars) (void)hipFree(h->d_scalars);
    if (h->h_pinned) (void)hipHostFree(h->h_pinned);
    if (h->d_flags) (void)hipFree(h->d_flags);
    if (h->d_trsv_gran) (void)hipFree(h->d_trsv_gran);
    if (h->d_loss_work) (void)hipFree(h->d_loss_work);
    if (h->d_trsv_gran2) (void)hipFree(h->d_trsv_gran2);
    for (int i = 0; i < 2; ++i) if (h->ev_loss[i]) (void)hipEventDestroy(h->ev_loss[i]);
    if (h->d_obflags) (void)hipFree(h->d_obflags);
    if (h->d_pts) (void)hipFree(h->d_pts);
    if (h->d_work) (void)hipFree(h->d_work);
    gpk_i_sk_free(h);
    if (h->d_splitk_ws) (void)hipFree(h->d_splitk_ws);
    if (h->d_splitk_cnt) (void)hipFree(h->d_splitk_cnt);
    for (hipEvent_t e : h->pipe_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->pipe_tev) (void)hipEventDestroy(e);
    if (h->pipe_g) (void)hipStreamDestroy(h->pipe_g);
    if (h->pipe_c) (void)hipStreamDestroy(h->pipe_c);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (int i = 0; i < 3; ++i) {
        if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int i = 0; i < 5; ++i) if (h->pev[i]) (void)hipEventDestroy(h->pev[i]);
    for (int i = 0; i < 2; ++i) if (h->asm_ev[i]) (void)hipEventDestroy(h->asm_ev[i]);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return 0;
}

extern "C" const char* gpk_last_error(gpk_handle h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int gpk_set_stream(gpk_handle h, void* s) {
    if (!h) return GPK_ERR_ARG;
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return 0;
}

extern "C" int gpk_synchronize(gpk_handle h) {
    if (!h) return GPK_ERR_ARG;
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int gpk_device_info(gpk_handle h, char* name, int name_len, int* cus, size_t* hbm, int* clock_khz) {
    if (!h) return GPK_ERR_ARG;
    hipDeviceProp_t prop;
    GPK_HIP(h, hipGetDeviceProperties(&prop, h->device));
    if (name && name_len > 0) {                       // (prop.name is EMPTY on some boxes of the pool: the architecture string alone then)
        if (prop.name[0]) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
        else snprintf(name, name_len, "%s", prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = prop.totalGlobalMem;
    if (clock_khz) *clock_khz = prop.clockRate;
    return 0;
}

extern "C" int gpk_malloc(gpk_handle h, size_t bytes, void** dptr) {
    if (!h || !dptr) return GPK_ERR_ARG;
    GPK_HIP(h, hipSetDevice(h->device));
    GPK_HIP(h, hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}

extern "C" int gpk_free(gpk_handle h, void* dptr) {
    if (!h) return GPK_ERR_ARG;
    if (!dptr) return 0;
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    GPK_HIP(h, hipFree(dptr));
    return 0;
}

extern "C" int gpk_memset(gpk_handle h, void* dptr, int value, size_t bytes) {
    if (!h) return GPK_ERR_ARG;
    GPK_HIP(h, hipMemsetAsync(dptr, value, bytes, h->stream));
    return 0;
}

// Host buffers handed over by ctypes are pageable: the copies are synchronous w.r.t. the host buffer.
extern "C" int gpk_memcpy_h2d(gpk_handle h, void* dst, const void* src, size_t bytes) {
    if (!h) return GPK_ERR_ARG;
    GPK_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int gpk_memcpy_d2h(gpk_handle h, void* dst, const void* src, size_t bytes) {
    if (!h) return GPK_ERR_ARG;
    GPK_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int gpk_memcpy_d2d(gpk_handle h, void* dst, const void* src, size_t bytes) {
    if (!h) return GPK_ERR_ARG;
    GPK_HIP(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}
extern "C" int gpk_memcpy2d_h2d(gpk_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    if (!h) return GPK_ERR_ARG;
    if (width == 0 || height == 0) return 0;
    GPK_HIP(h, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int gpk_memcpy2d_d2h(gpk_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    if (!h) return GPK_ERR_ARG;
    if (width == 0 || height == 0) return 0;
    GPK_HIP(h, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}
extern "C" int gpk_memcpy2d_d2d(gpk_handle h, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    if (!h) return GPK_ERR_ARG;
    if (width == 0 || height == 0) return 0;
    GPK_HIP(h, hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

extern "C" int gpk_timer_start(gpk_handle h) {
    if (!h) return GPK_ERR_ARG;
    GPK_HIP(h, hipEventRecord(h->ev0, h->stream));
    return 0;
}
extern "C" int gpk_timer_stop(gpk_handle h, double* ms) {
    if (!h || !ms) return GPK_ERR_ARG;
    GPK_HIP(h, hipEventRecord(h->ev1, h->stream));
    GPK_HIP(h, hipEventSynchronize(h->ev1));
    float f = 0.f;
    GPK_HIP(h, hipEventElapsedTime(&f, h->ev0, h->ev1));
    *ms = (double)f;
    return 0;
}

// Per-handle development / tuning switch (include/gpk_debug.h): one field of the handle's GpkTune (gpk_common.h, where every key is
// documented next to its default).  No process-wide state: two handles of one process can run different variants side by side.
extern "C" int gpk_tune(gpk_handle h, int key, int value) {
    if (!h) return GPK_ERR_ARG;
    GpkTune& t = h->tune;
    // values that selected a superseded design (potf2 + row solve as two launches, persistent outer-block kernel, first / third panel kernel,
    // flag-chained single-vector solve): removed in round 6 -- they had lost every comparison since round 2 (git 4be18eb holds them)
    if ((key == 5 && value == 0) || (key == 7 && value != 0) || (key == 21 && value != 1) || (key == 4 && value == 2))
        return gpk_bad_arg(h, "gpk_tune: this variant was removed in round 6");
#ifndef GPK_DEV
    if (key == 11 || (key == 54 && value != 0))
        return gpk_bad_arg(h, "gpk_tune: this variant exists only in the development build (libgpk_dev.so)");
#endif
    switch (key) {
        case 0: t.force_cfg = value; return 0;
        case 2: t.mt_trsm = value; return 0;
        case 3: t.strip = value; return 0;
        case 4: t.fused_trsv = value; return 0;
        case 5: t.fused_panel = value; return 0;
        case 6: t.supertile = value; return 0;
        case 7: t.persistent_ob = value; return 0;
        case 8: t.k64_small = value; return 0;
        case 9: t.gemm_extra_lds = value; return 0;
        case 10: t.use_dinv = value; return 0;
        case 11: t.probe_chain_cus = value; return 0;
        case 12: t.pipeline = value; return 0;
        case 13: t.pipeline_chain_cus = value; return 0;
        case 14: t.pipeline_max_n = value; return 0;
        case 15: t.stagger = value; return 0;
        case 16: t.rev_k = value; return 0;
        case 17: t.pipeline_pre = value; return 0;
        case 18: t.left_looking_panels = value; return 0;
        case 19: t.potrf_pipeline_min_n = value; return 0;
        case 20: t.potrf_pipeline_max_n = value; return 0;
        case 21: t.panel_mfma = value; return 0;
        case 23: t.eikonal_lz = value; return 0;
        case 24: t.pipeline_units = value; return 0;
        case 25: t.force_splitk = value; return 0;
        case 26: t.pipeline_lookahead = value; return 0;
        case 28: t.pipeline_w0 = value; return 0;
        case 29: t.pipeline_ob = value; return 0;
        case 30: t.solve_splitk = value; return 0;
        case 33: t.tall_min = value; return 0;
        case 34: t.pipeline_tile = value; return 0;
        case 35: t.band_mb = value; return 0;
        case 36: t.syrk_band = value; return 0;
        case 38: t.big_min = value; return 0;
        case 40: t.structured = value; return 0;
        case 41: t.panel_unrolled = value; return 0;
        case 42: t.sk = value; return 0;
        case 43: t.sk_rounds = value; return 0;
        case 44: t.sk_snap = value; return 0;
        case 45: t.sk_stagger = value; return 0;
        case 46: t.sk_rowclass = value; return 0;
        case 47: t.asm_pairs = value; return 0;
        case 48: t.panel_fused = value; return 0;
        case 49: t.row_order = value; return 0;
        case 50: t.big_lower_min = value; return 0;
        case 51: t.potrf_ob = value; return 0;
        case 52: t.exact_loss = value; return 0;
        case 54: t.potrf_lookahead = value; return 0;
        case 56: t.seq_left_looking = value != 0; return 0;
        case 55: if (value < 0 || value > 3) return gpk_bad_arg(h, "gpk_tune: key 55 takes 0 .. 3"); t.asm_nt = value; return 0;
        default: return gpk_bad_arg(h, "gpk_tune: unknown key");
    }
}

extern "C" int gpk_prof_enable(gpk_handle h, int on) {
    if (!h) return GPK_ERR_ARG;
    if (on && !h->pev[0]) for (int i = 0; i < 5; ++i) GPK_HIP(h, hipEventCreate(&h->pev[i]));
    h->prof = on != 0;
    for (int i = 0; i < 4; ++i) h->prof_ms[i] = 0.0;
    h->prof_cnt = 0;
    h->prof_syrk_ms = 0.0;
    h->prof_pipelined = 0;
    h->prof_phase = 0;
    for (int i = 0; i < 4; ++i) { h->prof_flops[i] = 0.0; h->prof_launches[i] = 0; }
    return 0;
}

extern "C" int gpk_prof_read_assembly(gpk_handle h, double* host_ms) {
    if (!h || !host_ms) return GPK_ERR_ARG;
    if (!h->asm_timed) return gpk_bad_arg(h, "prof_read_assembly: no gpk_assemble call was timed (gpk_prof_enable first)");
    GPK_HIP(h, hipEventSynchronize(h->asm_ev[1]));
    float ms = 0.f;
    GPK_HIP(h, hipEventElapsedTime(&ms, h->asm_ev[0], h->asm_ev[1]));
    *host_ms = (double)ms;
    return 0;
}

extern "C" int gpk_prof_read_flops(gpk_handle h, double* host_flops4, long* host_launches4) {
    if (!h || !host_flops4) return GPK_ERR_ARG;
    for (int i = 0; i < 4; ++i) {
        host_flops4[i] = h->prof_flops[i];
        if (host_launches4) host_launches4[i] = h->prof_launches[i];
    }
    return 0;
}

extern "C" int gpk_prof_read_pipeline(gpk_handle h, int* host_pipelined, double* host_syrk_launch_ms, int* host_chain_cus) {
    if (!h) return GPK_ERR_ARG;
    if (host_pipelined) *host_pipelined = h->prof_pipelined;
    if (host_syrk_launch_ms) *host_syrk_launch_ms = h->prof_syrk_ms;
    if (host_chain_cus) *host_chain_cus = h->pipe_chain_cus;
    return 0;
}

extern "C" int gpk_prof_read(gpk_handle h, double* host_ms4, int* host_count) {
    if (!h || !host_ms4 || !host_count) return GPK_ERR_ARG;
    for (int i = 0; i < 4; ++i) host_ms4[i] = h->prof_ms[i];
    *host_count = h->prof_cnt;
    return 0;
}

int gpk_i_ensure_points(gpk_handle h, size_t doubles) {
    if (h->pts_cap >= doubles) return 0;
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    if (h->d_pts) GPK_HIP(h, hipFree(h->d_pts));
    h->d_pts = nullptr; h->pts_cap = 0;
    GPK_HIP(h, hipMalloc((void**)&h->d_pts, doubles * sizeof(double)));
    h->pts_cap = doubles;
    return 0;
}

// Reserved on first use (the first pipelined or forced split-K launch of a handle), each buffer on its own: a handle that never
// splits a product holds neither, and a failed second allocation does not orphan the first.
int gpk_i_splitk_reserve(gpk_handle h) {
    constexpr size_t WS = (size_t)64 << 20;
    constexpr int NCNT = 65536;
    if (!h->d_splitk_cnt) {
        GPK_HIP(h, hipMalloc(&h->d_splitk_cnt, NCNT * sizeof(unsigned)));
        hipError_t e = hipMemset(h->d_splitk_cnt, 0, NCNT * sizeof(unsigned));
        if (e != hipSuccess) { (void)hipFree(h->d_splitk_cnt); h->d_splitk_cnt = nullptr; return gpk_fail(h, e, "hipMemset", __FILE__, __LINE__); }
        h->splitk_cnt_cap = NCNT;
    }
    if (!h->d_splitk_ws) {
        GPK_HIP(h, hipMalloc(&h->d_splitk_ws, WS));
        h->splitk_ws_cap = WS;
    }
    return 0;
}

int gpk_i_workspace(gpk_handle h, size_t bytes, double** out) {
    if (h->work_cap < bytes) {
        GPK_HIP(h, hipStreamSynchronize(h->stream));
        if (h->d_work) GPK_HIP(h, hipFree(h->d_work));
        h->d_work = nullptr; h->work_cap = 0;
        for (long& v : h->work_sig) v = 0;
        GPK_HIP(h, hipMalloc((void**)&h->d_work, bytes));
        h->work_cap = bytes;
    }
    *out = h->d_work;
    return 0;
}

// gpk_mg.hip -- multi-GPU (one process per GPU) schedule of the factor/solve path behind the C ABI (include/gpk_mg.h).
//
// Replaces, for the >= 10k-point configuration, the same reference work as gpk_potrf / gpk_gn_step: jnp.linalg.cholesky of
// the nugget-regularised Gram matrix (reference src/PDEs.py:75-80) and the Hessian_GN / GN_method step (src/PDEs.py:89-127),
// which the reference can only run on one device (README.md:9).
//
// Everything numerical is the single-GPU building blocks of this library (gpk_i_potrf_panel, gpk_i_gemm,
// gpk_i_trsm_left_dinv, ...); this file holds (1) the SCHEDULE of the panel-sharded Cholesky as a plan -- a flat list of
// operations on three streams with explicit event dependencies, produced by a pure host function so that CPU tests can
// interpret it -- (2) its executor on HIP streams / events with the collectives reached through ncclBroadcast /
// ncclAllGather-shaped function pointers, (3) the column-sharded Gauss-Newton step, (4) the dlopen() binding of RCCL.
#include "gpk_common.h"
#include "../../include/gpk_mg.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>

struct gpk_mg_ctx {
    gpk_handle h = nullptr;
    int rank = 0, world = 1, nb = 512;
    int lookahead = 0, shard_hb = 0, col_align = 128;
    int overlap_s = -1;                    // step: 0 = one all-gather of the column shards of S (padded to the widest), 1 = one broadcast per shard
                                           // (exact sizes) on the communication stream with the block-row products of Hb chasing the arrivals,
                                           // -1 (default) = by the bytes each form puts on a link: the shards are cut by WORK, so their widths
                                           // differ (widest / mean = 1.30 / 1.57 / 1.79 at 2 / 4 / 8 ranks of config 5) and the all-gather moves
                                           // (P - 1) x widest columns per rank against nc columns for the broadcasts (gpk_mg_set_option key 3);
                                           // 2 = DIRECT exchange (round 6): every rank sends its shard to every peer and receives theirs inside
                                           // one ncclGroupStart / ncclGroupEnd -- exact sizes, and on xGMI (point-to-point links, one per
                                           // peer) the P - 1 transfers of a rank travel over P - 1 different links at once instead of
                                           // around a ring; the all-gather of the block rows of Hb takes the same form
    void* comm = nullptr;
    gpk_mg_bcast_fn bcast = nullptr;
    gpk_mg_allgather_fn allgather = nullptr;
    gpk_mg_send_fn p2p_send = nullptr;     // ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd (optional: the direct exchange, key 3 = 2)
    gpk_mg_recv_fn p2p_recv = nullptr;
    gpk_mg_group_fn p2p_begin = nullptr, p2p_end = nullptr;
    int p2p_off = 0;                       // gpk_mg_set_option key 4 = 0: behave as if the point-to-point entry points were not bound
    bool own_comm = false;
    void* lib = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    const char* (*errstr)(int) = nullptr;
    hipStream_t s_panel = nullptr, s_comm = nullptr;
    std::vector<hipEvent_t> ev;
    hipEvent_t ev_begin = nullptr;
    double* buf[2] = {nullptr, nullptr};   // transfer buffers of the panel broadcasts (contiguous (n - k0) x kb images)
    size_t buf_cap = 0;                    // bytes each
    double* gsend = nullptr; double* grecv = nullptr;   // persistent all-gather staging of the step
    size_t gsend_cap = 0, grecv_cap = 0;
    int* d_infos = nullptr;                // world + 1 ints: [0, world) gathered, [world] mine
    std::vector<int> bounds;
};

namespace {

constexpr int NCCL_INT32 = 2, NCCL_DOUBLE = 8;

struct Op { int kind, a, b, stream; };

void make_plan(int n, int nb, int P, int r, int lookahead, std::vector<Op>& ops) {
    ops.clear();
    if (n <= 0) return;
    const int nblk = gpk_ceil_div(n, nb);
    auto own = [&](int k) { return k % P == r; };
    auto EA = [&](int k) { return k; };                 // panel k in place on this rank
    auto EC = [&](int k) { return nblk + k; };          // main stream applied panel k to my next look-ahead column
    auto EB = [&](int k) { return 2 * nblk + k; };      // transfer buffer of panel k free again
    if (!lookahead) {
        // one stream, no events: factor -> (pack, broadcast, unpack) -> update my columns, panel after panel
        for (int k = 0; k < nblk; ++k) {
            if (own(k)) { ops.push_back({GPK_MG_FACTOR, k, 0, 0}); if (P > 1) ops.push_back({GPK_MG_PACK, k, 0, 0}); }
            if (P > 1) {
                ops.push_back({GPK_MG_BCAST, k, k % P, 0});
                if (!own(k)) ops.push_back({GPK_MG_UNPACK, k, 0, 0});
            }
            for (int j = k + 1; j < nblk; ++j) if (own(j)) ops.push_back({GPK_MG_UPDATE, j, k, 0});
        }
        return;
    }
    constexpr int M = 0, PN = 1, C = 2;
    if (own(0)) {
        ops.push_back({GPK_MG_FACTOR, 0, 0, PN});
        if (P > 1) ops.push_back({GPK_MG_PACK, 0, 0, PN});
        ops.push_back({GPK_MG_RECORD, EA(0), 0, PN});
    }
    for (int k = 0; k < nblk; ++k) {
        // ---- panel k travels (communication stream)
        if (P > 1) {
            if (own(k)) {
                ops.push_back({GPK_MG_WAIT, EA(k), 0, C});
                ops.push_back({GPK_MG_BCAST, k, k % P, C});
            } else {
                ops.push_back({GPK_MG_BCAST, k, k % P, C});
                ops.push_back({GPK_MG_UNPACK, k, 0, C});
                ops.push_back({GPK_MG_RECORD, EA(k), 0, C});
            }
            ops.push_back({GPK_MG_RECORD, EB(k), 0, C});
        }
        // ---- look-ahead (panel stream): the owner of panel k+1 applies panel k to that block column FIRST and factors it,
        //      so that its broadcast overlaps with everybody's remaining updates by panel k
        if (k + 1 < nblk && own(k + 1)) {
            ops.push_back({GPK_MG_WAIT, EA(k), 0, PN});
            if (k >= 1) ops.push_back({GPK_MG_WAIT, EC(k - 1), 0, PN});   // column k+1 has received panels 0 .. k-1 on the main stream
            ops.push_back({GPK_MG_UPDATE, k + 1, k, PN});
            ops.push_back({GPK_MG_FACTOR, k + 1, 0, PN});
            if (P > 1) {
                if (k >= 1) ops.push_back({GPK_MG_WAIT, EB(k - 1), 0, PN});   // slot (k+1) mod 2 last carried panel k-1
                ops.push_back({GPK_MG_PACK, k + 1, 0, PN});
            }
            ops.push_back({GPK_MG_RECORD, EA(k + 1), 0, PN});
        }
        // ---- trailing updates of my other columns (main stream), nearest column first: it is my next look-ahead column
        bool first = true;
        for (int j = k + 2; j < nblk; ++j) {
            if (!own(j)) continue;
            if (first) ops.push_back({GPK_MG_WAIT, EA(k), 0, M});
            ops.push_back({GPK_MG_UPDATE, j, k, M});
            if (first) ops.push_back({GPK_MG_RECORD, EC(k), 0, M});
            first = false;
        }
    }
    // join: the caller's stream continues only when the last panel is in place and the communication stream has drained
    ops.push_back({GPK_MG_WAIT, EA(nblk - 1), 0, M});
    if (P > 1) ops.push_back({GPK_MG_WAIT, EB(nblk - 1), 0, M});
}

// the same rule for any leading-zero layout: column c costs (rows - first_row(c))^2 (round 6: the Burgers and Eikonal systems)
template <class FirstRow>
void column_bounds_by(int ncols, int rows, int P, int align, FirstRow first_row, std::vector<int>& b) {
    b.assign(1, 0);
    if (align < 1) align = 1;
    std::vector<double> w((size_t)std::max(ncols, 0));
    double acc = 0.0;
    for (int c = 0; c < ncols; ++c) {
        const double len = (double)rows - (double)std::max(0, first_row(c));
        acc += len * len;
        w[c] = acc;
    }
    for (int r = 1; r < P; ++r) {
        int cut = 0;
        if (ncols > 0) {
            const double target = w[ncols - 1] * r / P;
            cut = (int)(std::lower_bound(w.begin(), w.end(), target) - w.begin());
        }
        cut = gpk_ceil_div(cut, align) * align;
        cut = std::min(std::max(cut, b.back()), ncols);
        b.push_back(cut);
    }
    b.push_back(ncols);
}

void column_bounds(int ncols, int lead, int rows, int P, int align, std::vector<int>& b) {
    // mirrors gpk/sharded.py::column_ranges_lz (numpy cumsum + searchsorted(side='left')); all values are integers < 2^53
    b.assign(1, 0);
    if (align < 1) align = 1;
    std::vector<double> w((size_t)std::max(ncols, 0));
    double acc = 0.0;
    for (int c = 0; c < ncols; ++c) {
        const double len = (double)rows - (double)std::max(0, lead - 1 - c);
        acc += len * len;
        w[c] = acc;
    }
    for (int r = 1; r < P; ++r) {
        int cut = 0;
        if (ncols > 0) {
            const double target = w[ncols - 1] * r / P;
            cut = (int)(std::lower_bound(w.begin(), w.end(), target) - w.begin());
        }
        cut = gpk_ceil_div(cut, align) * align;
        cut = std::min(std::max(cut, b.back()), ncols);
        b.push_back(cut);
    }
    b.push_back(ncols);
}

int nccl_fail(gpk_mg_handle mg, int r, const char* what) {
    char msg[256];
    snprintf(msg, sizeof msg, "%s failed: %s (ncclResult %d)", what, (mg && mg->errstr) ? mg->errstr(r) : "collective stand-in", r);
    if (mg && mg->h) mg->h->err = msg;
    return -(10000 + r);
}

#define MG_NCCL(mg, call, what) do { int r__ = (call); if (r__ != 0) return nccl_fail((mg), r__, (what)); } while (0)

int ensure_buffers(gpk_mg_handle mg, size_t panel_bytes) {
    gpk_handle h = mg->h;
    if (mg->buf_cap >= panel_bytes) return 0;
    GPK_HIP(h, hipDeviceSynchronize());
    for (int i = 0; i < 2; ++i) { if (mg->buf[i]) GPK_HIP(h, hipFree(mg->buf[i])); mg->buf[i] = nullptr; }
    mg->buf_cap = 0;
    for (int i = 0; i < 2; ++i) GPK_HIP(h, hipMalloc((void**)&mg->buf[i], panel_bytes));
    mg->buf_cap = panel_bytes;
    return 0;
}

// Exact-size all-to-all of one buffer per rank (the "v" form of an all-gather) through grouped point-to-point calls: rank r's cnt[r]
// doubles at `send` reach every peer's recv + off[r].  My own part is not moved.  Every rank issues the same group.
int exchange_direct(gpk_mg_handle mg, const double* send, double* recv, const std::vector<size_t>& cnt, const std::vector<size_t>& off,
                    hipStream_t s, const char* what) {
    if (!gpk_mg_has_p2p(mg))
        return gpk_bad_arg(mg->h, "gpk_mg: the direct exchange needs ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd (gpk_mg_set_p2p)");
    const int P = mg->world, me = mg->rank;
    MG_NCCL(mg, mg->p2p_begin(), what);
    int r1 = 0;
    for (int d = 1; d < P && r1 == 0; ++d) {                          // peers in "distance" order: rank me + d receives, rank me - d sends
        const int to = (me + d) % P, from = (me - d + P) % P;
        if (cnt[me] > 0) r1 = mg->p2p_send(send, cnt[me], NCCL_DOUBLE, to, mg->comm, (void*)s);
        if (r1 == 0 && cnt[from] > 0) r1 = mg->p2p_recv(recv + off[from], cnt[from], NCCL_DOUBLE, from, mg->comm, (void*)s);
    }
    const int r2 = mg->p2p_end();                                     // (always closed, also after a failed call inside the group)
    if (r1 != 0) return nccl_fail(mg, r1, what);
    if (r2 != 0) return nccl_fail(mg, r2, what);
    return 0;
}

int ensure_gather(gpk_mg_handle mg, size_t send_bytes, size_t recv_bytes) {
    gpk_handle h = mg->h;
    if (mg->gsend_cap < send_bytes) {
        GPK_HIP(h, hipDeviceSynchronize());
        if (mg->gsend) GPK_HIP(h, hipFree(mg->gsend));
        mg->gsend = nullptr; mg->gsend_cap = 0;
        GPK_HIP(h, hipMalloc((void**)&mg->gsend, send_bytes));
        mg->gsend_cap = send_bytes;
    }
    if (mg->grecv_cap < recv_bytes) {
        GPK_HIP(h, hipDeviceSynchronize());
        if (mg->grecv) GPK_HIP(h, hipFree(mg->grecv));
        mg->grecv = nullptr; mg->grecv_cap = 0;
        GPK_HIP(h, hipMalloc((void**)&mg->grecv, recv_bytes));
        mg->grecv_cap = recv_bytes;
    }
    return 0;
}

int ensure_streams(gpk_mg_handle mg, size_t nev) {
    gpk_handle h = mg->h;
    if (!mg->s_panel) {
        int lo = 0, hi = 0;
        GPK_HIP(h, hipDeviceGetStreamPriorityRange(&lo, &hi));      // (numerically lower = higher priority)
        GPK_HIP(h, hipStreamCreateWithPriority(&mg->s_panel, hipStreamNonBlocking, hi));
        GPK_HIP(h, hipStreamCreateWithPriority(&mg->s_comm, hipStreamNonBlocking, hi));
        GPK_HIP(h, hipEventCreateWithFlags(&mg->ev_begin, hipEventDisableTiming));
    }
    while (mg->ev.size() < nev) {
        hipEvent_t e;
        GPK_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        mg->ev.push_back(e);
    }
    return 0;
}

// The plan, executed: block operations through the handle (its stream switched per operation), collectives through the
// bound function pointers.  Nothing here synchronises with the host.  pivot indices are reported relative to A.
int exec_potrf(gpk_mg_handle mg, double* A, int n, int lda) {
    gpk_handle h = mg->h;
    const int nb = mg->nb, P = mg->world;
    if (n <= 0) return 0;
    const int nblk = gpk_ceil_div(n, nb);
    const int la = (mg->lookahead && nblk > 1) ? 1 : 0;
    std::vector<Op> plan;
    make_plan(n, nb, P, mg->rank, la, plan);
    if (P > 1) {
        if (!mg->bcast) return gpk_bad_arg(h, "gpk_mg: world > 1 needs a communicator (gpk_mg_rccl_init / gpk_mg_set_comm)");
        GPK_TRY(ensure_buffers(mg, (size_t)n * (size_t)std::min(nb, n) * sizeof(double)));
    }
    const hipStream_t main_s = h->stream;
    hipStream_t streams[3] = {main_s, main_s, main_s};
    if (la) {
        GPK_TRY(ensure_streams(mg, 3 * (size_t)nblk));
        streams[1] = mg->s_panel; streams[2] = mg->s_comm;
        GPK_HIP(h, hipEventRecord(mg->ev_begin, main_s));           // the matrix was produced on the caller's stream
        GPK_HIP(h, hipStreamWaitEvent(mg->s_panel, mg->ev_begin, 0));
        GPK_HIP(h, hipStreamWaitEvent(mg->s_comm, mg->ev_begin, 0));
    }
    int rc = 0;
    for (const Op& op : plan) {
        const hipStream_t s = streams[op.stream];
        h->stream = s;
        h->no_sk = (la && op.stream != 0) ? 1 : 0;                   // tile-list GEMM launches (one workspace per handle) only on the main stream
        const int k = (op.kind == GPK_MG_UPDATE) ? op.b : op.a;
        const int k0 = k * nb, kb = std::min(nb, n - k0);
        hipError_t e = hipSuccess;
        switch (op.kind) {
        case GPK_MG_FACTOR:
            rc = gpk_i_potrf_panel(h, A + (long)k0 * lda + k0, n - k0, kb, lda, k0);
            break;
        case GPK_MG_PACK:
            e = hipMemcpy2DAsync(mg->buf[k & 1], (size_t)kb * 8, A + (long)k0 * lda + k0, (size_t)lda * 8, (size_t)kb * 8, (size_t)(n - k0),
                                 hipMemcpyDeviceToDevice, s);
            break;
        case GPK_MG_UNPACK:
            e = hipMemcpy2DAsync(A + (long)k0 * lda + k0, (size_t)lda * 8, mg->buf[k & 1], (size_t)kb * 8, (size_t)kb * 8, (size_t)(n - k0),
                                 hipMemcpyDeviceToDevice, s);
            break;
        case GPK_MG_BCAST: {
            const int r = mg->bcast(mg->buf[k & 1], mg->buf[k & 1], (size_t)(n - k0) * kb, NCCL_DOUBLE, op.b, mg->comm, (void*)s);
            if (r != 0) rc = nccl_fail(mg, r, "panel broadcast");
            break;
        }
        case GPK_MG_UPDATE: {
            const int j0 = op.a * nb, jb = std::min(nb, n - j0);
            const double* Lj = A + (long)j0 * lda + k0;              // rows j0.. of panel k
            rc = gpk_i_gemm(h, false, true, n - j0, jb, kb, -1.0, Lj, lda, Lj, lda, 1.0, A + (long)j0 * lda + j0, lda, false, 0, false, true);
            break;
        }
        case GPK_MG_RECORD: e = hipEventRecord(mg->ev[op.a], s); break;
        case GPK_MG_WAIT:   e = hipStreamWaitEvent(s, mg->ev[op.a], 0); break;
        default: rc = gpk_bad_arg(h, "gpk_mg: corrupt plan");
        }
        if (e != hipSuccess) rc = gpk_fail(h, e, "gpk_mg plan operation", __FILE__, __LINE__);
        if (rc) break;
    }
    h->stream = main_s;
    h->no_sk = 0;
    if (rc && la) { (void)hipStreamSynchronize(mg->s_panel); (void)hipStreamSynchronize(mg->s_comm); }
    return rc;
}

// LAPACK info over all ranks after a sharded factorisation: every rank saw only the panels it factored.  One all-gather of
// one int; the smallest positive index wins (negative = a device-side wait expired somewhere).
int gather_info(gpk_mg_handle mg, int* host_info) {
    gpk_handle h = mg->h;
    const int P = mg->world;
    if (P == 1) {
        GPK_HIP(h, hipMemcpyAsync(host_info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        GPK_HIP(h, hipStreamSynchronize(h->stream));
        return 0;
    }
    GPK_HIP(h, hipMemcpyAsync(mg->d_infos + P, h->d_info, sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    MG_NCCL(mg, mg->allgather(mg->d_infos + P, mg->d_infos, 1, NCCL_INT32, mg->comm, (void*)h->stream), "all-gather of info");
    std::vector<int> all((size_t)P);
    GPK_HIP(h, hipMemcpyAsync(all.data(), mg->d_infos, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    GPK_HIP(h, hipStreamSynchronize(h->stream));
    int info = 0;
    for (int v : all) {
        if (v < 0) { info = v; break; }
        if (v > 0 && (info == 0 || v < info)) info = v;
    }
    *host_info = info;
    return 0;
}

struct nccl_unique_id { char internal[128]; };

// dlerror() returns its message ONCE and clears it: read it right after the failing dlopen, before the next attempt resets it
void* open_rccl(const char* path, std::string* err = nullptr) {
    void* lib = dlopen(path && path[0] ? path : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        const char* e = dlerror();
        if (err) *err = e ? e : "?";
        if (!(path && path[0])) {
            lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) { const char* e2 = dlerror(); if (err) *err += std::string("; ") + (e2 ? e2 : "?"); }
        }
    }
    return lib;
}

const char* const RCCL_SYMBOLS[] = {"ncclGetUniqueId", "ncclCommInitRank", "ncclBroadcast", "ncclAllGather"};

}  // namespace

extern "C" int gpk_mg_plan_potrf(int n, int nb, int world, int rank, int lookahead, int* host_ops, int cap, int* host_count) {
    if (n < 0 || nb <= 0 || world <= 0 || rank < 0 || rank >= world || !host_count) return GPK_ERR_ARG;
    std::vector<Op> ops;
    make_plan(n, nb, world, rank, (lookahead && gpk_ceil_div(n, nb) > 1) ? 1 : 0, ops);
    *host_count = (int)ops.size();
    if (host_ops) {
        const int m = std::min(cap, (int)ops.size());
        for (int i = 0; i < m; ++i) { host_ops[4 * i] = ops[i].kind; host_ops[4 * i + 1] = ops[i].a; host_ops[4 * i + 2] = ops[i].b; host_ops[4 * i + 3] = ops[i].stream; }
    }
    return 0;
}

extern "C" int gpk_mg_column_bounds(int ncols, int lead, int rows, int world, int align, int* host_bounds) {
    if (ncols < 0 || world <= 0 || !host_bounds) return GPK_ERR_ARG;
    std::vector<int> b;
    column_bounds(ncols, lead, rows, world, align, b);
    for (int i = 0; i <= world; ++i) host_bounds[i] = b[i];
    return 0;
}

extern "C" int gpk_mg_create(gpk_handle h, int rank, int world, int panel_width, gpk_mg_handle* out) {
    if (!h || !out || world <= 0 || rank < 0 || rank >= world) return GPK_ERR_ARG;
    if (panel_width <= 0 || panel_width > 512 || panel_width % 64 != 0) return gpk_bad_arg(h, "gpk_mg_create: panel width must be a multiple of 64, at most 512");
    gpk_mg_ctx* mg = new gpk_mg_ctx();
    mg->h = h; mg->rank = rank; mg->world = world; mg->nb = panel_width;
    mg->lookahead = world > 1 ? 1 : 0;
    mg->shard_hb = world >= 4 ? 1 : 0;
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipMalloc((void**)&mg->d_infos, (size_t)(world + 1) * sizeof(int));
    if (e != hipSuccess) { delete mg; return gpk_fail(h, e, "gpk_mg_create", __FILE__, __LINE__); }
    *out = mg;
    return 0;
}

extern "C" int gpk_mg_destroy(gpk_mg_handle mg) {
    if (!mg) return 0;
    gpk_handle h = mg->h;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (mg->own_comm && mg->comm && mg->comm_destroy) (void)mg->comm_destroy(mg->comm);
    for (hipEvent_t e : mg->ev) (void)hipEventDestroy(e);
    if (mg->ev_begin) (void)hipEventDestroy(mg->ev_begin);
    if (mg->s_panel) (void)hipStreamDestroy(mg->s_panel);
    if (mg->s_comm) (void)hipStreamDestroy(mg->s_comm);
    for (int i = 0; i < 2; ++i) if (mg->buf[i]) (void)hipFree(mg->buf[i]);
    if (mg->gsend) (void)hipFree(mg->gsend);
    if (mg->grecv) (void)hipFree(mg->grecv);
    if (mg->d_infos) (void)hipFree(mg->d_infos);
    delete mg;                                                        // (the library handle of dlopen stays: RCCL may hold threads)
    return 0;
}

extern "C" int gpk_mg_set_comm(gpk_mg_handle mg, void* comm, gpk_mg_bcast_fn bcast, gpk_mg_allgather_fn allgather) {
    if (!mg || !bcast || !allgather) return GPK_ERR_ARG;
    mg->comm = comm; mg->bcast = bcast; mg->allgather = allgather; mg->own_comm = false;
    return 0;
}

extern "C" int gpk_mg_set_p2p(gpk_mg_handle mg, gpk_mg_send_fn send, gpk_mg_recv_fn recv, gpk_mg_group_fn group_start, gpk_mg_group_fn group_end) {
    if (!mg || !send || !recv || !group_start || !group_end) return GPK_ERR_ARG;
    mg->p2p_send = send; mg->p2p_recv = recv; mg->p2p_begin = group_start; mg->p2p_end = group_end;
    return 0;
}

extern "C" int gpk_mg_has_p2p(gpk_mg_handle mg) { return (mg && !mg->p2p_off && mg->p2p_send && mg->p2p_recv && mg->p2p_begin && mg->p2p_end) ? 1 : 0; }

extern "C" int gpk_mg_set_option(gpk_mg_handle mg, int key, int value) {
    if (!mg) return GPK_ERR_ARG;
    if (key == 0) { mg->lookahead = value != 0; return 0; }
    if (key == 1) { mg->shard_hb = value != 0; return 0; }
    if (key == 2 && value >= 1) { mg->col_align = value; return 0; }
    if (key == 4) {                                                   // escape hatch: 0 = never use ncclSend / ncclRecv (falls back to the collectives)
        mg->p2p_off = value == 0;
        if (mg->p2p_off && mg->overlap_s == 2) mg->overlap_s = -1;
        return 0;
    }
    if (key == 3 && value >= -1 && value <= 2) {
        if (value == 2 && !gpk_mg_has_p2p(mg)) return gpk_bad_arg(mg->h, "gpk_mg_set_option: key 3 = 2 needs the point-to-point entry points (gpk_mg_set_p2p)");
        mg->overlap_s = value; return 0;
    }
    return gpk_bad_arg(mg->h, "gpk_mg_set_option: key / value");
}

// Can this process bind RCCL at all?  dlopen + the four entry points, nothing else (no communicator, no GPU call): lets every
// rank AGREE on the answer (an all-reduce over the caller's process group) BEFORE any of them enters ncclCommInitRank, where a
// rank whose peer never arrives would block.  errbuf (may be NULL) receives the reason.
extern "C" int gpk_mg_rccl_probe(const char* librccl_path, char* errbuf, int cap) {
    std::string err;
    void* lib = open_rccl(librccl_path, &err);
    if (lib) {
        for (const char* sym : RCCL_SYMBOLS)
            if (!dlsym(lib, sym)) { err = std::string(sym) + " not found in the library"; lib = nullptr; break; }
    }
    if (!lib && errbuf && cap > 0) snprintf(errbuf, (size_t)cap, "%s", err.c_str());
    return lib ? 0 : GPK_ERR_NODEV;
}

extern "C" int gpk_mg_rccl_unique_id(const char* librccl_path, void* host_id128) {
    if (!host_id128) return GPK_ERR_ARG;
    void* lib = open_rccl(librccl_path);
    if (!lib) return GPK_ERR_NODEV;
    auto get = (int (*)(nccl_unique_id*))dlsym(lib, "ncclGetUniqueId");
    if (!get) return GPK_ERR_NODEV;
    const int r = get((nccl_unique_id*)host_id128);
    return r == 0 ? 0 : -(10000 + r);
}

extern "C" int gpk_mg_rccl_init(gpk_mg_handle mg, const char* librccl_path, const void* host_id128) {
    if (!mg || !host_id128) return GPK_ERR_ARG;
    gpk_handle h = mg->h;
    std::string lerr;
    void* lib = open_rccl(librccl_path, &lerr);
    if (!lib) { h->err = "gpk_mg_rccl_init: cannot load RCCL: " + lerr; return GPK_ERR_NODEV; }
    auto init = (int (*)(void**, int, nccl_unique_id, int))dlsym(lib, "ncclCommInitRank");
    auto bc = (gpk_mg_bcast_fn)dlsym(lib, "ncclBroadcast");
    auto ag = (gpk_mg_allgather_fn)dlsym(lib, "ncclAllGather");
    mg->comm_destroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    mg->errstr = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    if (!init || !bc || !ag) { h->err = "gpk_mg_rccl_init: ncclCommInitRank / ncclBroadcast / ncclAllGather not found in the library"; return GPK_ERR_NODEV; }
    GPK_HIP(h, hipSetDevice(h->device));
    nccl_unique_id id;
    memcpy(&id, host_id128, sizeof id);
    void* comm = nullptr;
    MG_NCCL(mg, init(&comm, mg->world, id, mg->rank), "ncclCommInitRank");
    mg->lib = lib; mg->comm = comm; mg->bcast = bc; mg->allgather = ag; mg->own_comm = true;
    // optional: the point-to-point entry points of the same library (every RCCL has them; a stand-in library may not)
    auto sd = (gpk_mg_send_fn)dlsym(lib, "ncclSend");
    auto rv = (gpk_mg_recv_fn)dlsym(lib, "ncclRecv");
    auto g0 = (gpk_mg_group_fn)dlsym(lib, "ncclGroupStart");
    auto g1 = (gpk_mg_group_fn)dlsym(lib, "ncclGroupEnd");
    if (sd && rv && g0 && g1) { mg->p2p_send = sd; mg->p2p_recv = rv; mg->p2p_begin = g0; mg->p2p_end = g1; }
    return 0;
}

// Round trip through the two bound collectives on small buffers: a broadcast from rank 0 and an all-gather, both checked on the
// host.  With a communicator of one rank (all a one-GPU box can have) this is what proves that the dlsym'ed entry points are
// called with the right argument order and data-type codes; with more ranks it is a cheap health check before the big buffers.
extern "C" int gpk_mg_selftest(gpk_mg_handle mg, int* host_ok) {
    if (!mg || !host_ok) return GPK_ERR_ARG;
    gpk_handle h = mg->h;
    *host_ok = 0;
    if (!mg->bcast || !mg->allgather) return gpk_bad_arg(h, "gpk_mg_selftest: no communicator bound");
    const int P = mg->world, n = 1000;
    const bool p2p = gpk_mg_has_p2p(mg) != 0;
    double* d = nullptr;
    GPK_HIP(h, hipMalloc((void**)&d, (size_t)n * (P + 3) * sizeof(double)));
    std::vector<double> host((size_t)n * (P + 3), -7.0);
    for (int i = 0; i < n; ++i) { host[i] = (mg->rank == 0) ? 1000.0 + i : -1.0; host[n + i] = 100.0 * mg->rank + 0.001 * i; }
    int rc = 0;
    hipError_t e = hipMemcpyAsync(d, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        int r = mg->bcast(d, d, (size_t)n, NCCL_DOUBLE, 0, mg->comm, (void*)h->stream);
        if (r == 0) r = mg->allgather(d + n, d + 2 * n, (size_t)n, NCCL_DOUBLE, mg->comm, (void*)h->stream);
        if (r != 0) rc = nccl_fail(mg, r, "self-test collective");
    }
    if (rc == 0 && e == hipSuccess && p2p) {
        // (round 6) the point-to-point entry points: one group in which every rank sends its pattern to its right neighbour and receives the
        // left neighbour's (one rank: to and from itself) -- argument order and type codes of ncclSend / ncclRecv, group start / end
        const int to = (mg->rank + 1) % P, from = (mg->rank + P - 1) % P;
        int r = mg->p2p_begin();
        int r1 = 0;
        if (r == 0) {
            r1 = mg->p2p_send(d + n, (size_t)n, NCCL_DOUBLE, to, mg->comm, (void*)h->stream);
            if (r1 == 0) r1 = mg->p2p_recv(d + (size_t)(P + 2) * n, (size_t)n, NCCL_DOUBLE, from, mg->comm, (void*)h->stream);
            r = mg->p2p_end();
        }
        if (r1 != 0 || r != 0) rc = nccl_fail(mg, r1 ? r1 : r, "self-test send / recv");
    }
    if (rc == 0 && e == hipSuccess) e = hipMemcpyAsync(host.data(), d, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (rc == 0 && e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return gpk_fail(h, e, "gpk_mg_selftest", __FILE__, __LINE__);
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) ok = host[i] == 1000.0 + i;
    for (int r = 0; r < P && ok; ++r)
        for (int i = 0; i < n && ok; ++i) ok = host[(size_t)(2 + r) * n + i] == 100.0 * r + 0.001 * i;
    if (p2p) {
        const int from = (mg->rank + P - 1) % P;
        for (int i = 0; i < n && ok; ++i) ok = host[(size_t)(P + 2) * n + i] == 100.0 * from + 0.001 * i;
    }
    *host_ok = ok ? 1 : 0;
    return 0;
}

// Bandwidth preflight of the bound collectives on buffers of the size the schedule moves (bench.py, before the first timed run on a
// fabric the code has never seen): `reps` broadcasts of `bytes` from every root in turn and `reps` all-gathers of bytes / world per
// rank, each after one untimed warm-up call, timed with HIP events on the handle's stream.  host_bcast_ms[world]: average
// milliseconds of one broadcast per root; *host_allgather_ms: average of one all-gather; *host_ranks_seen: how many DISTINCT ranks
// the all-gather delivered (each rank contributes its rank id) -- must equal world.
extern "C" int gpk_mg_preflight(gpk_mg_handle mg, size_t bytes, int reps, double* host_bcast_ms, double* host_allgather_ms, int* host_ranks_seen) {
    if (!mg || !host_bcast_ms || !host_allgather_ms || !host_ranks_seen || reps < 1 || bytes < 8) return GPK_ERR_ARG;
    gpk_handle h = mg->h;
    if (!mg->bcast || !mg->allgather) return gpk_bad_arg(h, "gpk_mg_preflight: no communicator bound");
    const int P = mg->world;
    const size_t n = bytes / sizeof(double), per = std::max<size_t>(n / P, 1);
    double* d = nullptr;
    GPK_HIP(h, hipMalloc((void**)&d, (n + per * (P + 1)) * sizeof(double)));
    double* gs = d + n;
    double* gr = gs + per;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0, (n + per * (P + 1)) * sizeof(double), h->stream);
    const double me = (double)mg->rank;
    if (e == hipSuccess) e = hipMemcpyAsync(gs, &me, sizeof(double), hipMemcpyHostToDevice, h->stream);
    for (int root = 0; root < P && rc == 0 && e == hipSuccess; ++root) {
        int r = mg->bcast(d, d, n, NCCL_DOUBLE, root, mg->comm, (void*)h->stream);            // warm-up (connection set-up of this root)
        if (r == 0) e = hipEventRecord(e0, h->stream);
        for (int i = 0; i < reps && r == 0; ++i) r = mg->bcast(d, d, n, NCCL_DOUBLE, root, mg->comm, (void*)h->stream);
        if (r != 0) { rc = nccl_fail(mg, r, "preflight broadcast"); break; }
        if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        host_bcast_ms[root] = (double)ms / reps;
    }
    if (rc == 0 && e == hipSuccess) {
        int r = mg->allgather(gs, gr, per, NCCL_DOUBLE, mg->comm, (void*)h->stream);
        if (r == 0) e = hipEventRecord(e0, h->stream);
        for (int i = 0; i < reps && r == 0; ++i) r = mg->allgather(gs, gr, per, NCCL_DOUBLE, mg->comm, (void*)h->stream);
        if (r != 0) rc = nccl_fail(mg, r, "preflight all-gather");
        if (rc == 0 && e == hipSuccess) e = hipEventRecord(e1, h->stream);
        if (rc == 0 && e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (rc == 0 && e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        *host_allgather_ms = (double)ms / reps;
    }
    int seen = 0;
    if (rc == 0 && e == hipSuccess) {
        std::vector<char> mark(P, 0);
        for (int r = 0; r < P && e == hipSuccess; ++r) {
            double v = -1.0;
            e = hipMemcpy(&v, gr + (size_t)r * per, sizeof(double), hipMemcpyDeviceToHost);
            const int id = (int)v;
            if (e == hipSuccess && v == (double)id && id >= 0 && id < P && !mark[id]) { mark[id] = 1; ++seen; }
        }
    }
    *host_ranks_seen = seen;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return gpk_fail(h, e, "gpk_mg_preflight", __FILE__, __LINE__);
    return 0;
}

// The same for the direct exchange (gpk_mg_set_option key 3 = 2): `reps` grouped exchanges in which every rank sends bytes / world to every
// peer and receives as much from each, after one untimed warm-up.  *host_ms: average milliseconds of one exchange.
extern "C" int gpk_mg_preflight_p2p(gpk_mg_handle mg, size_t bytes, int reps, double* host_ms) {
    if (!mg || !host_ms || reps < 1 || bytes < 8) return GPK_ERR_ARG;
    gpk_handle h = mg->h;
    if (!gpk_mg_has_p2p(mg)) return gpk_bad_arg(h, "gpk_mg_preflight_p2p: no point-to-point entry points bound");
    const int P = mg->world;
    const size_t per = std::max<size_t>(bytes / sizeof(double) / P, 1);
    double* d = nullptr;
    GPK_HIP(h, hipMalloc((void**)&d, per * (size_t)(P + 1) * sizeof(double)));
    std::vector<size_t> cnt((size_t)P, per), off((size_t)P);
    for (int r = 0; r < P; ++r) off[r] = (size_t)r * per;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemsetAsync(d, 0, per * (size_t)(P + 1) * sizeof(double), h->stream);
    int rc = e == hipSuccess ? exchange_direct(mg, d + per * P, d, cnt, off, h->stream, "preflight direct exchange") : 0;   // warm-up
    if (rc == 0 && e == hipSuccess) e = hipEventRecord(e0, h->stream);
    for (int i = 0; i < reps && rc == 0 && e == hipSuccess; ++i) rc = exchange_direct(mg, d + per * P, d, cnt, off, h->stream, "preflight direct exchange");
    if (rc == 0 && e == hipSuccess) e = hipEventRecord(e1, h->stream);
    if (rc == 0 && e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (rc == 0 && e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    *host_ms = (double)ms / reps;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return gpk_fail(h, e, "gpk_mg_preflight_p2p", __FILE__, __LINE__);
    return 0;
}

extern "C" int gpk_mg_potrf(gpk_mg_handle mg, double* A, int n, int lda, int* host_info) {
    if (!mg || !A || n < 0 || lda < n) return GPK_ERR_ARG;
    gpk_handle h = mg->h;
    GPK_HIP(h, hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
    // one rank, no look-ahead requested: the single-GPU routine (one lower-triangular trailing update per panel instead of one
    // launch per block column: 253 vs 293 ms at order 34000, tools/mg_lookahead_probe.py); the plan is for more than one rank
    if (mg->world == 1 && !mg->lookahead) GPK_TRY(gpk_i_potrf(h, A, n, lda, 0));
    else GPK_TRY(exec_potrf(mg, A, n, lda));
    if (host_info) GPK_TRY(gather_info(mg, host_info));
    return 0;
}

extern "C" int gpk_mg_gn_step(gpk_mg_handle mg, const gpk_gn_problem* p, double* z, double step_size, double* S, int lds, double* S2,
                              double* Hb, int ldh, double* delta, double* host_loss_in, int* host_info) {
    if (!mg || !p || !z || !S || !Hb || !delta) return GPK_ERR_ARG;
    gpk_handle h = mg->h;
    const int P = mg->world, rank = mg->rank, nb = mg->nb;
    if (P == 1) return gpk_gn_step(h, p, z, step_size, S, lds, Hb, ldh, delta, host_loss_in, host_info);
    // round 6: the Eikonal and Burgers systems as well (one factor each), in the staircase orders of their gpk_gn_step, and the Darcy system
    // with its cached a-part (gpk_gn_darcy_prepare: only the u-part is sharded -- columns of L_u^{-1}A_u cut by work under its piecewise
    // profile; the a-part contributes the constant H_a and one replicated F column).  The relaxed system is one-GPU only.
    const int rev = gpk_i_gn_layout(h, p);
    const bool darcy = p->system == GPK_GN_DARCY;
    if ((p->system != GPK_GN_ELLIPTIC && p->system != GPK_GN_EIKONAL && p->system != GPK_GN_BURGERS && !darcy) || rev < 1 || rev > 4 || (darcy != (rev == 4)))
        return gpk_bad_arg(h, "gpk_mg_gn_step: elliptic, Eikonal, Burgers and Darcy systems only (leading-zero layout)");
    if (!p->Dinv || !S2) return gpk_bad_arg(h, "gpk_mg_gn_step: needs the inverted diagonal blocks (Dinv) and S2");
    if (darcy && (!p->Wa || !p->Ha || !p->Dinv2)) return gpk_bad_arg(h, "gpk_mg_gn_step: the Darcy system needs its cached a-part (gpk_gn_darcy_prepare) and Dinv2");
    struct LayoutGuard {                                             // Eikonal: its profile in the handle; Burgers: the slope; reset on every way out
        gpk_handle h; LayoutGuard(gpk_handle hh, const gpk_gn_problem* pp, int r) : h(hh) { gpk_i_gn_layout_enter(h, pp, r); }
        ~LayoutGuard() { gpk_i_gn_layout_leave(h); }
    } layout_guard(h, p, rev);
    if (!mg->allgather) return gpk_bad_arg(h, "gpk_mg: world > 1 needs a communicator (gpk_mg_rccl_init / gpk_mg_set_comm)");
    int nz = 0, rows = 0;
    GPK_TRY(gpk_i_gn_dims(h, p, &nz, &rows));
    const int nc = nz + 1;
    if (lds < nc || ldh < nc) return gpk_bad_arg(h, "gpk_mg_gn_step: lds/ldh < nz+1");
    const int db = p->dinv_block > 0 ? p->dinv_block : 256;
    const hipStream_t s = h->stream;
    // ---- the loss of the iterate the step starts from, by true substitution (replicated: one vector; as in gpk_gn_step, gpk_tune key 52)
    double* d_exact = nullptr;
    if (h->tune.exact_loss) GPK_TRY(gpk_i_gn_exact_loss(h, p, z, &d_exact));
    // ---- S <- [A | F] in the leading-zero layout on every rank (a memset + O(N)); my column shard of L^{-1}[A | F] -> S2
    GPK_TRY(gpk_gn_build_rev(h, p, z, S, lds));
    // Darcy: rows [0, na) = a-part (factor L2, cached), [na, na + nu) = u-part (factor L: the sharded solve), then the data rows (no factor).
    // Everything below works on the rows from `xoff` on: xrows of them are exchanged (the solved u-part), prows enter the products.
    const int na = darcy ? 3 * p->Nd : 0, nu = darcy ? 4 * p->Nd + p->Nb : rows;
    const int xoff = na, xrows = nu, prows = rows - na;
    double* const Sx = S + (long)xoff * lds;
    double* const S2x = S2 + (long)xoff * lds;
    if (darcy) gpk_i_gn_darcy_profile(h, p->Nd);                      // the u-part's three-segment profile into the handle (reset by the guard)
    if (rev == 1) column_bounds(nc, nz, rows, P, mg->col_align, mg->bounds);
    else column_bounds_by(nc, xrows, P, mg->col_align, [&](int c) { return gpk_i_gn_first_row(h, nz, c); }, mg->bounds);
    const std::vector<int>& b = mg->bounds;
    const int c0 = b[rank], c1 = b[rank + 1];
    int per = 0;
    for (int r = 0; r < P; ++r) per = std::max(per, b[r + 1] - b[r]);
    if (c1 > c0) {
        h->stair_base = c0;                                          // (piecewise profile: my shard's column 0 in the profile's frame)
        const int rc = gpk_i_trsm_left_dinv(h, p->L, p->Dinv, db, xrows, p->ldl, Sx + c0, lds, S2x + c0, lds, c1 - c0, darcy ? 1 : std::max(nz - c0, 0), 0);
        h->stair_base = 0; h->stair_col0 = h->stair_row0 = 0;        // (the products below read S2 from its column 0, row 0)
        GPK_TRY(rc);
    }
    if (darcy) {
        // replicated on every rank (tiny): the data rows (identity factor: all columns) and the a-part's F column L_a^{-1}[w1; w2; w0]
        if (p->Ndata > 0)
            GPK_HIP(h, hipMemcpy2DAsync(S2x + (long)nu * lds, (size_t)lds * 8, Sx + (long)nu * lds, (size_t)lds * 8, (size_t)nc * 8, (size_t)p->Ndata,
                                        hipMemcpyDeviceToDevice, h->stream));
        GpkStair keep = h->stair;                                    // (a dense one-column solve: no profile)
        h->stair = GpkStair();
        const int rc = gpk_i_trsm_left_dinv(h, p->L2, p->Dinv2, db, na, p->ldl2, S + nz, lds, S2 + nz, lds, 1, 0, 0);
        h->stair = keep; h->stair_col0 = h->stair_row0 = 0;
        GPK_TRY(rc);
    }
    // ---- exchange of the column shards of S2, then my block rows (cyclic) of the lower triangle of Hb = S2^T S2 (structural zeros
    //      skipped).  Two forms of the exchange (gpk_mg_set_option key 3; bench.py times both on the fabric and keeps the faster):
    //      0: ONE all-gather (shards padded to the widest; persistent staging buffers), products afterwards;
    //      1: one BROADCAST per shard, in rank order, on the communication stream (exact sizes, no padding); block row i only needs the
    //         columns [0, i0 + ib), i.e. the shards up to the one that holds its last column -- its product is issued behind that
    //         shard's event, so the early block rows are computed while the later shards are still travelling.
    const size_t shard = (size_t)xrows * per;
    const bool direct = mg->overlap_s == 2;                           // exact sizes, grouped point-to-point (round 6)
    const bool overlap_s = direct ? true : mg->overlap_s < 0 ? (long)(P - 1) * per > (long)nc : mg->overlap_s != 0;   // (a function of the shapes: all ranks agree; `true` also selects the exact-size packing)
    const int nblk = gpk_ceil_div(nc, nb), per_rank = gpk_ceil_div(nblk, P);
    // the block rows travel as their LOWER parts only (round 4): block row i is ib x (i0 + ib); every rank's share padded to the largest
    std::vector<size_t> share((size_t)P, 0);
    for (int i = 0; i < nblk; ++i) { const int i0 = i * nb, ib = std::min(nb, nc - i0); share[i % P] += (size_t)ib * (size_t)(i0 + ib); }
    const size_t hshare = *std::max_element(share.begin(), share.end());
    (void)per_rank;
    const size_t all_s = overlap_s ? (size_t)xrows * nc : shard * P;
    GPK_TRY(ensure_gather(mg, std::max(shard, hshare) * sizeof(double), std::max(all_s, hshare * P) * sizeof(double)));
    if (c1 > c0)
        GPK_HIP(h, hipMemcpy2DAsync(mg->gsend, (size_t)(overlap_s ? c1 - c0 : per) * 8, S2x + c0, (size_t)lds * 8, (size_t)(c1 - c0) * 8, (size_t)xrows,
                                    hipMemcpyDeviceToDevice, s));
    auto block_row = [&](int i) {
        const int i0 = i * nb, ib = std::min(nb, nc - i0);
        h->stair_col0 = h->stair_row0 = 0;
        int rc = gpk_i_gemm(h, true, false, ib, i0 + ib, prows, 1.0, S2x + i0, lds, S2x, lds, 0.0, Hb + (long)i0 * ldh, ldh, false, darcy ? 1 : nz);
        // Darcy: + the cached a-part -- H_a on the rows of this block that lie in [N_d, 4 N_d), and in the block that holds the border row its
        // a-part terms (L_a^{-1}F_a)^T W_a and |L_a^{-1}F_a|^2, exactly the launches of the one-GPU step
        if (rc == 0 && darcy) rc = gpk_i_gn_darcy_add_a(h, p, Hb, ldh, i0, i0 + ib, S2 + nz, lds);
        return rc;
    };
    if (direct) {
        std::vector<size_t> cnt((size_t)P), off((size_t)P);
        size_t o = 0;
        for (int r = 0; r < P; ++r) { cnt[r] = (size_t)xrows * (size_t)(b[r + 1] - b[r]); off[r] = o; o += cnt[r]; }
        GPK_TRY(exchange_direct(mg, mg->gsend, mg->grecv, cnt, off, s, "direct exchange of the shards of S"));
        for (int r = 0; r < P; ++r) {
            const int a0 = b[r], a1 = b[r + 1];
            if (r == rank || a1 <= a0) continue;
            GPK_HIP(h, hipMemcpy2DAsync(S2x + a0, (size_t)lds * 8, mg->grecv + off[r], (size_t)(a1 - a0) * 8, (size_t)(a1 - a0) * 8, (size_t)xrows,
                                        hipMemcpyDeviceToDevice, s));
        }
        for (int i = rank; i < nblk; i += P) GPK_TRY(block_row(i));
    } else if (!overlap_s) {
        MG_NCCL(mg, mg->allgather(mg->gsend, mg->grecv, shard, NCCL_DOUBLE, mg->comm, (void*)s), "all-gather of S");
        for (int r = 0; r < P; ++r) {
            const int a0 = b[r], a1 = b[r + 1];
            if (r == rank || a1 <= a0) continue;
            GPK_HIP(h, hipMemcpy2DAsync(S2x + a0, (size_t)lds * 8, mg->grecv + (size_t)r * shard, (size_t)per * 8, (size_t)(a1 - a0) * 8, (size_t)xrows,
                                        hipMemcpyDeviceToDevice, s));
        }
        for (int i = rank; i < nblk; i += P) GPK_TRY(block_row(i));
    } else {
        GPK_TRY(ensure_streams(mg, (size_t)P + 1));
        const hipStream_t cs = mg->s_comm;
        GPK_HIP(h, hipEventRecord(mg->ev_begin, s));                 // my shard is solved and packed
        GPK_HIP(h, hipStreamWaitEvent(cs, mg->ev_begin, 0));
        size_t off = 0;
        for (int r = 0; r < P; ++r) {                                // every rank issues the same broadcasts in the same order
            const int a0 = b[r], a1 = b[r + 1];
            const size_t cnt = (size_t)xrows * (size_t)(a1 - a0);
            if (cnt > 0) {
                double* dst = mg->grecv + off;
                MG_NCCL(mg, mg->bcast(r == rank ? mg->gsend : dst, dst, cnt, NCCL_DOUBLE, r, mg->comm, (void*)cs), "broadcast of a shard of S");
                if (r != rank)
                    GPK_HIP(h, hipMemcpy2DAsync(S2x + a0, (size_t)lds * 8, dst, (size_t)(a1 - a0) * 8, (size_t)(a1 - a0) * 8, (size_t)xrows,
                                                hipMemcpyDeviceToDevice, cs));
            }
            GPK_HIP(h, hipEventRecord(mg->ev[r], cs));
            off += cnt;
        }
        int waited = -1;                                             // shards 0 .. waited are known to the main stream
        for (int i = rank; i < nblk; i += P) {
            const int last_col = std::min(i * nb + nb, nc) - 1;
            int need = 0;
            while (need + 1 < P && b[need + 1] <= last_col) ++need;  // the shard that holds column last_col
            if (need > waited) { GPK_HIP(h, hipStreamWaitEvent(s, mg->ev[need], 0)); waited = need; }
            GPK_TRY(block_row(i));
        }
        if (waited < P - 1) GPK_HIP(h, hipStreamWaitEvent(s, mg->ev[P - 1], 0));   // join: the next collective of this communicator runs on s
    }
    // ---- all-gather of the block rows, lower parts only (block row i: ib rows of i0 + ib entries, packed)
    {
        size_t o = 0;
        for (int i = rank; i < nblk; i += P) {
            const int i0 = i * nb, ib = std::min(nb, nc - i0);
            GPK_HIP(h, hipMemcpy2DAsync(mg->gsend + o, (size_t)(i0 + ib) * 8, Hb + (long)i0 * ldh, (size_t)ldh * 8, (size_t)(i0 + ib) * 8, (size_t)ib,
                                        hipMemcpyDeviceToDevice, s));
            o += (size_t)ib * (size_t)(i0 + ib);
        }
        std::vector<size_t> hoff((size_t)P);
        if (direct) {                                                // exact shares, one link per peer
            size_t acc = 0;
            for (int r = 0; r < P; ++r) { hoff[r] = acc; acc += share[r]; }
            GPK_TRY(exchange_direct(mg, mg->gsend, mg->grecv, share, hoff, s, "direct exchange of the block rows of Hb"));
        } else {
            for (int r = 0; r < P; ++r) hoff[r] = (size_t)r * hshare;
            MG_NCCL(mg, mg->allgather(mg->gsend, mg->grecv, hshare, NCCL_DOUBLE, mg->comm, (void*)s), "all-gather of Hb");
        }
        for (int r = 0; r < P; ++r) {
            if (r == rank) continue;
            size_t oo = 0;
            for (int i = r; i < nblk; i += P) {
                const int i0 = i * nb, ib = std::min(nb, nc - i0);
                GPK_HIP(h, hipMemcpy2DAsync(Hb + (long)i0 * ldh, (size_t)ldh * 8, mg->grecv + hoff[r] + oo, (size_t)(i0 + ib) * 8,
                                            (size_t)(i0 + ib) * 8, (size_t)ib, hipMemcpyDeviceToDevice, s));
                oo += (size_t)ib * (size_t)(i0 + ib);
            }
        }
    }
    double* d_loss = h->d_scalars;
    GPK_HIP(h, hipMemcpyAsync(d_loss, Hb + (long)nz * ldh + nz, sizeof(double), hipMemcpyDeviceToDevice, s));
    // ---- Cholesky of the bordered matrix: replicated (every rank factors its own copy, no communication), or panel-sharded
    //      with the plan of the big factorisation; the last row of the factor is (L_H^{-1} g/2)^T either way
    GPK_HIP(h, hipMemsetAsync(h->d_info, 0, sizeof(int), s));
    if (mg->shard_hb) GPK_TRY(exec_potrf(mg, Hb, nc, ldh));
    else GPK_TRY(gpk_i_potrf(h, Hb, nc, ldh, 0));
    GPK_TRY(gpk_i_gn_finish(h, p, nz, rev, Hb, ldh, S, delta, z, step_size));
    double loss = 0.0;
    int info = 0;
    GPK_HIP(h, hipMemcpyAsync(&loss, d_exact ? d_exact : d_loss, sizeof(double), hipMemcpyDeviceToHost, s));
    if (mg->shard_hb) GPK_TRY(gather_info(mg, &info));
    else {
        GPK_HIP(h, hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
        GPK_HIP(h, hipStreamSynchronize(s));
    }
    if (info == nc) info = 0;                                         // the border pivot is not part of H
    if (host_info) *host_info = info;
    if (host_loss_in) *host_loss_in = loss;
    return 0;
}

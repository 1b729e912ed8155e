/* gpk_mg.h -- multi-GPU entry points of libgpk.so: the panel-sharded Cholesky factorisation and the column-sharded
 * Gauss-Newton step of the >= 10k-point configuration (BASELINE config 5), ONE PROCESS PER GPU, exchanges over RCCL.
 *
 * What it replaces in the reference: the same calls as gpk_potrf / gpk_gn_step (jnp.linalg.cholesky, src/PDEs.py:77; the
 * Hessian_GN / GN_method step, src/PDEs.py:89-127) -- the reference has no multi-device path at all (README.md:9 concedes
 * the dense matrices are "costly when you use more than 10k collocation points"); this is SURVEY 8(e) / 8(b) B3 `gpk_mg_*`.
 *
 * Process model.  SURVEY B3 sketched a single-process ncclCommInitAll; the build runs one process per GPU (the launch
 * contract of bench.py and torch.distributed), so every process creates a gpk_handle on its own device and then a
 * gpk_mg_handle with its rank and the world size.  The collectives are reached through two function pointers that have
 * EXACTLY the signatures of ncclBroadcast and ncclAllGather:
 *   - production: gpk_mg_rccl_init() dlopen()s the RCCL library the process already uses (torch's bundled librccl.so, or
 *     /opt/rocm/lib/librccl.so.1), creates a communicator from a 128-byte ncclUniqueId that rank 0 obtained with
 *     gpk_mg_rccl_unique_id() and shipped to the other ranks by any means (bench.py: the torch.distributed store), and
 *     binds ncclBroadcast / ncclAllGather; or gpk_mg_set_comm() with an existing ncclComm_t passed as an opaque pointer;
 *   - tests on a one-GPU box (several ranks share the device, which RCCL refuses): host-staged stand-ins.
 *
 * Schedule.  Cholesky: 1-D block-cyclic block columns of width `panel_width`; the owner factors its
 * tall panel with the fused panel kernels, packs it into a contiguous buffer, broadcasts it; every rank unpacks it into
 * its copy of the matrix (all ranks end with the full factor: 9.2 GB of 288 GB at config 5) and updates the block columns
 * it owns.  With look-ahead (default for world > 1) the owner of panel k+1 applies panel k to that block column and
 * factors it on a high-priority stream as soon as panel k has arrived, and its broadcast travels on a third stream while
 * all ranks are still applying panel k to the rest of their columns.  The schedule is a flat list of operations with
 * explicit event dependencies -- gpk_mg_plan_potrf() returns it, a pure host function that tests interpret on CPU.
 * Gauss-Newton step: column shards of S = L^{-1}[A | F] cut by work (leading-zero layout), all-gather of S, cyclic block
 * rows of Hb = S^T S, all-gather, Cholesky of Hb replicated (world <= 2) or panel-sharded with the same plan, replicated
 * triangular solve + update: every rank holds the same iterate bit for bit.
 *
 * Conventions: as gpk.h (return codes, device pointers, row-major double, asynchronous on the handle's stream except where
 * a host scalar is returned).  <0 return values from the collectives: -(10000 + ncclResult_t), text in gpk_last_error().
 */
#ifndef GPK_MG_H
#define GPK_MG_H

#include "gpk.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpk_mg_ctx* gpk_mg_handle;

/* signatures of ncclBroadcast / ncclAllGather (datatype: ncclDataType_t as int, ncclDouble = 8, ncclInt32 = 2;
 * comm: ncclComm_t; stream: hipStream_t); return 0 (ncclSuccess) or an ncclResult_t */
typedef int (*gpk_mg_bcast_fn)(const void* sendbuf, void* recvbuf, size_t count, int datatype, int root, void* comm, void* stream);
typedef int (*gpk_mg_allgather_fn)(const void* sendbuf, void* recvbuf, size_t sendcount, int datatype, void* comm, void* stream);

/* signatures of ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd: the DIRECT exchange of the step (round 6, key 3 = 2 below) */
typedef int (*gpk_mg_send_fn)(const void* sendbuf, size_t count, int datatype, int peer, void* comm, void* stream);
typedef int (*gpk_mg_recv_fn)(void* recvbuf, size_t count, int datatype, int peer, void* comm, void* stream);
typedef int (*gpk_mg_group_fn)(void);

/* h: this process' handle (its device, its stream = the stream results are ordered on).  panel_width: multiple of 64, <= 512. */
int gpk_mg_create(gpk_handle h, int rank, int world, int panel_width, gpk_mg_handle* out);
int gpk_mg_destroy(gpk_mg_handle mg);                          /* also destroys a communicator created by gpk_mg_rccl_init */
/* an existing communicator + the two collectives of the RCCL build that created it (or stand-ins) */
int gpk_mg_set_comm(gpk_mg_handle mg, void* comm, gpk_mg_bcast_fn bcast, gpk_mg_allgather_fn allgather);
/* optional: the four point-to-point entry points of the same library (or stand-ins); gpk_mg_rccl_init binds ncclSend / ncclRecv /
 * ncclGroupStart / ncclGroupEnd by itself when the library has them.  gpk_mg_has_p2p: 1 when they are bound. */
int gpk_mg_set_p2p(gpk_mg_handle mg, gpk_mg_send_fn send, gpk_mg_recv_fn recv, gpk_mg_group_fn group_start, gpk_mg_group_fn group_end);
int gpk_mg_has_p2p(gpk_mg_handle mg);
/* librccl_path: NULL = "librccl.so.1" by the loader's search order.  host_id128: 128 bytes. */
/* gpk_mg_rccl_probe: can this process load the library and find ncclGetUniqueId / ncclCommInitRank / ncclBroadcast /
 * ncclAllGather?  No communicator, no GPU call -- so that all ranks can agree on the answer BEFORE any of them enters
 * ncclCommInitRank (which blocks until every rank has arrived).  0 = yes; errbuf (may be NULL) receives the reason otherwise. */
int gpk_mg_rccl_probe(const char* librccl_path, char* errbuf, int errbuf_cap);
int gpk_mg_rccl_unique_id(const char* librccl_path, void* host_id128);
int gpk_mg_rccl_init(gpk_mg_handle mg, const char* librccl_path, const void* host_id128);
/* key 0: look-ahead (0 off, 1 on; default on for world > 1);  key 1: Cholesky of Hb (0 replicated, 1 panel-sharded;
 * default sharded for world >= 4);  key 2: alignment of the column shards of the step (default 128);
 * key 3: exchange of the column shards of S in the step (0 = one all-gather, every shard padded to the widest; 1 = one broadcast
 * per shard, exact sizes, on the communication stream, the block-row products of Hb issued behind the arrival of the shards they
 * read; -1 = default: the form that puts fewer bytes on a link -- broadcasts when (world - 1) x widest shard > all columns, which
 * holds from 4 ranks on for the work-balanced shards of BASELINE config 5 (widest / mean 1.57 at 4 ranks, 1.79 at 8);
 * 2 = DIRECT exchange: inside one ncclGroupStart / ncclGroupEnd every rank sends its shard to every peer and receives theirs (exact
 * sizes).  xGMI is point-to-point -- one link per peer -- so the world - 1 transfers of a rank use world - 1 links at once where a
 * ring collective is bound by one; the block rows of Hb are exchanged the same way.  Needs gpk_mg_has_p2p; bench.py times all forms);
 * key 4: 0 = never use the point-to-point entry points even when they are bound (gpk_mg_has_p2p then answers 0; GPK_MG_P2P=0 in the Python layer) */
int gpk_mg_set_option(gpk_mg_handle mg, int key, int value);

/* Health check of the bound collectives (every rank calls it): a broadcast from rank 0 and an all-gather of small buffers,
 * verified on the host; *host_ok = 1 if both delivered what they should. */
int gpk_mg_selftest(gpk_mg_handle mg, int* host_ok);

/* Bandwidth preflight of the bound collectives (every rank calls it, before the first large run on an unknown fabric): `reps`
 * broadcasts of `bytes` from every root in turn and `reps` all-gathers of bytes / world per rank, each after one untimed warm-up,
 * timed with HIP events.  host_bcast_ms[world]: average ms per broadcast and root; *host_allgather_ms: average ms per all-gather;
 * *host_ranks_seen: distinct ranks the all-gather delivered (== world when the communicator spans the job). */
int gpk_mg_preflight(gpk_mg_handle mg, size_t bytes, int reps, double* host_bcast_ms, double* host_allgather_ms, int* host_ranks_seen);
/* the same for the direct exchange: every rank sends bytes / world to every peer and receives as much from each; *host_ms per exchange */
int gpk_mg_preflight_p2p(gpk_mg_handle mg, size_t bytes, int reps, double* host_ms);

/* gpk_potrf over all ranks: A (n x n, ld lda) holds the SAME symmetric matrix on every rank on entry and the complete
 * lower factor on every rank on return.  host_info: LAPACK info, identical on all ranks (one host read at the end). */
int gpk_mg_potrf(gpk_mg_handle mg, double* A, int n, int lda, int* host_info);

/* gpk_gn_step over all ranks (elliptic system and, since round 6, the Eikonal and Burgers systems -- one factor each, the same column
 * shards cut by work under their own leading-zero profiles -- and the Darcy system, which must carry its cached a-part (Wa / Ha of
 * gpk_gn_darcy_prepare) and Dinv2: its u-part is sharded, the a-part and the data rows are replicated; the relaxed system: one GPU only; host_prob->L = the replicated factor, host_prob->Dinv/dinv_block = its
 * inverted diagonal blocks, gpk_trtri_diag).  S, S2: s_rows x lds each; S2 must be ZERO before the first step and is then
 * reused across steps (the solve never writes left of the leading-zero boundary).  z is updated identically on every rank.
 * world == 1: the call is gpk_gn_step itself (S2 unused), bit for bit. */
int gpk_mg_gn_step(gpk_mg_handle mg, const gpk_gn_problem* host_prob, double* z, double step_size, double* S, int lds, double* S2,
                   double* Hb, int ldh, double* delta, double* host_loss_in, int* host_info);

/* ---- the schedule as data (pure host functions: no device, no handle) -------------------------------------------------- */
/* operation kinds of a plan entry {kind, a, b, stream}; streams: 0 = the handle's stream (trailing updates), 1 = panel
 * stream (high priority), 2 = communication stream.  Events are numbered 0 .. 3*nblk-1:
 *   k           panel k is in place in A on this rank (factored here, or received and unpacked)
 *   nblk + k    the main stream has applied panel k to this rank's next look-ahead column
 *   2 nblk + k  the transfer buffer of panel k (slot k mod 2) is free again */
enum { GPK_MG_FACTOR = 0,   /* a = k: factor block column k (rows k*nb .. n) in place                                   */
       GPK_MG_PACK = 1,     /* a = k: block column k -> transfer buffer k mod 2                                          */
       GPK_MG_BCAST = 2,    /* a = k, b = root rank: broadcast transfer buffer k mod 2                                   */
       GPK_MG_UNPACK = 3,   /* a = k: transfer buffer k mod 2 -> block column k                                          */
       GPK_MG_UPDATE = 4,   /* a = j, b = k: A[j0:, j-block] -= A[j0:, k-block] A[j-block rows, k-block]^T               */
       GPK_MG_RECORD = 5,   /* a = event                                                                                 */
       GPK_MG_WAIT = 6 };   /* a = event                                                                                 */
/* host_ops: cap entries of 4 ints; *host_count = entries needed (call with cap = 0 to size).  Every WAIT refers to an event
 * RECORDed earlier in the list, so issuing the list in order is a valid host order for any stream mapping. */
int gpk_mg_plan_potrf(int n, int nb, int world, int rank, int lookahead, int* host_ops, int cap, int* host_count);
/* work-balanced contiguous column shards of the leading-zero right-hand side (column c < lead starts at row lead-1-c, its
 * solve costs ~(rows - start)^2): host_bounds[world + 1], multiples of `align` except the last */
int gpk_mg_column_bounds(int ncols, int lead, int rows, int world, int align, int* host_bounds);

#ifdef __cplusplus
}
#endif
#endif /* GPK_MG_H */

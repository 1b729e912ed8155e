"""Binding of include/gpk_mg.h: the native multi-GPU schedule (one process per GPU) -- panel-sharded Cholesky with look-ahead
and the column-sharded Gauss-Newton step, RCCL reached from C through dlopen.

Python's part is plumbing only: it ships the 128-byte ncclUniqueId from rank 0 to the other ranks (torch.distributed, any
backend) and hands raw device pointers to the library.  On a one-GPU test box, where RCCL refuses several ranks on one
device, `comm='staged'` binds host-staged stand-ins for ncclBroadcast / ncclAllGather (torch.distributed over gloo) to the
same entry points, so that the native schedule itself runs unchanged.

`plan_potrf` / `column_bounds` are pure host functions of the library (no GPU needed): the schedule as data.
"""
import ctypes as C
import os

import numpy as np

from ._lib import GpkError, MG_ALLGATHER_FN, MG_BCAST_FN, MG_GROUP_FN, MG_RECV_FN, MG_SEND_FN, load_library

OP_NAMES = {0: 'FACTOR', 1: 'PACK', 2: 'BCAST', 3: 'UNPACK', 4: 'UPDATE', 5: 'RECORD', 6: 'WAIT'}
FACTOR, PACK, BCAST, UNPACK, UPDATE, RECORD, WAIT = range(7)
_DTYPE_BYTES = {0: 1, 1: 1, 2: 4, 3: 4, 4: 8, 5: 8, 6: 2, 7: 4, 8: 8}        # ncclDataType_t -> bytes


def plan_potrf(n, nb, world, rank, lookahead=True):
    """The schedule of the panel-sharded Cholesky for one rank: list of (kind, a, b, stream) -- see gpk_mg.h."""
    lib = load_library()
    cnt = C.c_int()
    rc = lib.gpk_mg_plan_potrf(int(n), int(nb), int(world), int(rank), int(bool(lookahead)), None, 0, C.byref(cnt))
    if rc != 0:
        raise GpkError(f'gpk_mg_plan_potrf: invalid arguments ({rc})')
    buf = (C.c_int * (4 * max(cnt.value, 1)))()
    lib.gpk_mg_plan_potrf(int(n), int(nb), int(world), int(rank), int(bool(lookahead)), buf, cnt.value, C.byref(cnt))
    return [tuple(buf[4 * i:4 * i + 4]) for i in range(cnt.value)]


def column_bounds(ncols, lead, rows, world, align=128):
    """Work-balanced contiguous column shards of the leading-zero right-hand side (world + 1 boundaries)."""
    lib = load_library()
    out = (C.c_int * (world + 1))()
    rc = lib.gpk_mg_column_bounds(int(ncols), int(lead), int(rows), int(world), int(align), out)
    if rc != 0:
        raise GpkError(f'gpk_mg_column_bounds: invalid arguments ({rc})')
    return list(out)


def loaded_hip_runtime():
    """path of the libamdhip64 this process has mapped (None if none yet)"""
    try:
        for line in open('/proc/self/maps'):
            if 'libamdhip64.so' in line:
                return line.split()[-1]
    except OSError:
        pass
    return None


def torch_rccl_path():
    """The RCCL library that belongs to the ROCm stack this process ACTUALLY runs on.  torch wheels bundle their own
    libamdhip64 / libhsa-runtime64 / librccl under torch/lib; whichever of torch and libgpk.so is loaded first decides which
    HIP runtime the process uses (same soname), and RCCL opens "its" HSA runtime by file name -- a librccl from the other tree
    finds an uninitialised second copy and ncclCommInitRank fails with 'no ROCm-capable device is detected'.  So: the librccl
    next to the mapped libamdhip64 (torch/lib when torch was imported first, as in bench.py; /opt/rocm/lib otherwise)."""
    hip = loaded_hip_runtime()
    cands = []
    if hip:
        d = os.path.dirname(hip)
        cands += [os.path.join(d, 'librccl.so'), os.path.join(d, 'librccl.so.1')]
    else:
        try:
            import torch
            cands.append(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'))
        except Exception:
            pass
    cands += ['/opt/rocm/lib/librccl.so.1', '/opt/rocm/lib/librccl.so']
    for p in cands:
        if os.path.exists(p):
            return p
    return None


class MultiGpu:
    """gpk_mg_handle of this rank.  comm: 'rccl' (own communicator, unique id exchanged over torch.distributed), 'staged'
    (host-staged collectives over torch.distributed -- tests on one GPU), or None for world == 1."""

    def __init__(self, ctx, rank=0, world=1, panel=512, comm=None, group=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.rank, self.world, self.panel = int(rank), int(world), int(panel)
        h = C.c_void_p()
        ctx._chk(self.lib.gpk_mg_create(ctx.h, self.rank, self.world, self.panel, C.byref(h)))
        self.h = h
        self._keep = []
        try:
            if comm == 'rccl':                                     # (also with one rank when asked for: the binding and the self-test run)
                self._init_rccl(group)
            elif comm == 'staged':
                self._init_staged(group)
            elif self.world > 1:
                raise ValueError("world > 1 needs comm='rccl' or comm='staged'")
            if os.environ.get('GPK_MG_P2P', '1') == '0':             # escape hatch: never touch ncclSend / ncclRecv (direct exchange and its preflight off)
                self.set_option('p2p', 0)
        except BaseException:
            # the object is never handed to the caller: release the native handle (and a half-made communicator) before re-raising
            # (_init_rccl raises on EVERY rank by design when any rank cannot bind RCCL; a caller's fall-back must not leak it)
            self.close()
            raise

    # ---- communicators ------------------------------------------------------------------------------------------------
    def _init_rccl(self, group):
        """Failure-symmetric over the ranks (advisor, round 3): every step that can fail on ONE rank before ncclCommInitRank -- loading
        the library, finding its entry points, rank 0 obtaining the unique id -- is followed by an agreement over the caller's process
        group, and only when every rank is ready does any of them enter ncclCommInitRank (which blocks until all have arrived).
        On failure EVERY rank raises GpkError, so a caller's fallback takes the same branch everywhere."""
        import torch.distributed as dist
        have_pg = dist.is_available() and dist.is_initialized()
        if self.world > 1 and not have_pg:
            raise GpkError("comm='rccl' with world > 1 needs an initialised torch.distributed process group to ship the unique id")

        def agree(err):
            """err: this rank's error text or ''.  Returns the first error over all ranks ('' if none); collective."""
            if not have_pg:
                return err
            box = [None] * dist.get_world_size(group)
            dist.all_gather_object(box, err, group=group)
            return next((f'rank {r}: {e}' for r, e in enumerate(box) if e), '')

        path = torch_rccl_path()
        pb = path.encode() if path else None
        ebuf = C.create_string_buffer(512)
        rc = self.lib.gpk_mg_rccl_probe(pb, ebuf, 512)
        err = agree('' if rc == 0 else f'cannot bind RCCL ({path}): {ebuf.value.decode(errors="replace")}')
        if err:
            raise GpkError('gpk_mg: ' + err)
        uid = (C.c_char * 128)()
        uerr = ''
        if self.rank == 0:
            rc = self.lib.gpk_mg_rccl_unique_id(pb, uid)
            if rc != 0:
                uerr = f'gpk_mg_rccl_unique_id failed ({rc}); library {path}'
        # rank 0 ALWAYS takes part in the broadcast and ships either the id or an error marker
        box = [(('ERR', uerr) if uerr else bytes(uid.raw)) if self.rank == 0 else None]
        if have_pg:
            dist.broadcast_object_list(box, src=0, group=group)   # 128 bytes through the process group's own transport
        if isinstance(box[0], tuple):
            raise GpkError('gpk_mg: rank 0: ' + box[0][1])
        uid2 = (C.c_char * 128).from_buffer_copy(box[0])
        self.ctx._chk(self.lib.gpk_mg_rccl_init(self.h, pb, uid2))
        self.comm_kind = f'rccl ({path})'
        if not self.selftest():                                    # argument order / data-type codes of the bound entry points
            raise GpkError('gpk_mg: the RCCL collectives did not deliver the self-test pattern')

    def _init_staged(self, group):
        """ncclBroadcast / ncclAllGather stand-ins: wait for the stream, stage through host memory, torch.distributed (gloo)."""
        import torch
        import torch.distributed as dist
        hip = C.CDLL(loaded_hip_runtime() or 'libamdhip64.so')   # the runtime this process has mapped, not whatever the search path yields
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rank, world = self.rank, self.world

        def bcast(send, recv, count, dtype, root, comm, stream):
            try:
                nbytes = count * _DTYPE_BYTES[dtype]
                if hip.hipStreamSynchronize(stream) != 0:
                    return 1
                host = np.empty(nbytes, dtype=np.uint8)
                if rank == root and hip.hipMemcpy(host.ctypes.data, send, nbytes, 2) != 0:
                    return 1
                t = torch.from_numpy(host)
                dist.broadcast(t, src=root, group=group)
                if rank != root and hip.hipMemcpy(recv, host.ctypes.data, nbytes, 1) != 0:
                    return 1
                return 0
            except Exception:                                     # noqa: BLE001 -- a C caller cannot take an exception
                return 3

        def allgather(send, recv, count, dtype, comm, stream):
            try:
                nbytes = count * _DTYPE_BYTES[dtype]
                if hip.hipStreamSynchronize(stream) != 0:
                    return 1
                host = np.empty(nbytes, dtype=np.uint8)
                if hip.hipMemcpy(host.ctypes.data, send, nbytes, 2) != 0:
                    return 1
                outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(outs, torch.from_numpy(host), group=group)
                allb = torch.cat(outs).numpy()
                if hip.hipMemcpy(recv, allb.ctypes.data, nbytes * world, 1) != 0:
                    return 1
                return 0
            except Exception:                                     # noqa: BLE001
                return 3

        # ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd stand-ins (the direct exchange, set_option('overlap_s', 2)): calls are
        # queued between group start and group end and carried out there -- wait for the streams, stage the send buffers through host
        # memory, one isend / irecv per queued call (per pair of ranks at most one of each per group: matched in order), copy the
        # received buffers back
        pending = []

        def group_start():
            pending.clear()
            pending.append('open')
            return 0

        def send(buf, count, dtype, peer, comm, stream):
            if not pending:
                return 5                                          # (only inside a group here)
            pending.append(('s', buf, count * _DTYPE_BYTES[dtype], peer, stream))
            return 0

        def recv(buf, count, dtype, peer, comm, stream):
            if not pending:
                return 5
            pending.append(('r', buf, count * _DTYPE_BYTES[dtype], peer, stream))
            return 0

        def group_end():
            try:
                ops = [o for o in pending if o != 'open']
                pending.clear()
                for st in {o[4] for o in ops}:
                    if hip.hipStreamSynchronize(st) != 0:
                        return 1
                reqs, recvs, cache = [], [], {}
                # (a rank talking to itself -- the self-test with one rank: a local copy, gloo has no send-to-self)
                selfs = [o for o in ops if o[3] == rank]
                ops = [o for o in ops if o[3] != rank]
                for snd in [o for o in selfs if o[0] == 's']:
                    for rcv in [o for o in selfs if o[0] == 'r' and o[2] == snd[2]]:
                        if hip.hipMemcpy(rcv[1], snd[1], snd[2], 3) != 0:
                            return 1
                for kind, buf, nbytes, peer, _ in ops:
                    if kind == 's':
                        key = (buf, nbytes)
                        if key not in cache:                      # (the same shard goes to every peer: staged once)
                            host = np.empty(nbytes, dtype=np.uint8)
                            if hip.hipMemcpy(host.ctypes.data, buf, nbytes, 2) != 0:
                                return 1
                            cache[key] = torch.from_numpy(host)
                        reqs.append(dist.isend(cache[key], dst=dist.get_global_rank(group, peer) if group is not None else peer, group=group))
                    else:
                        t = torch.empty(nbytes, dtype=torch.uint8)
                        reqs.append(dist.irecv(t, src=dist.get_global_rank(group, peer) if group is not None else peer, group=group))
                        recvs.append((buf, t, nbytes))
                for q in reqs:
                    q.wait()
                for buf, t, nbytes in recvs:
                    if hip.hipMemcpy(buf, t.numpy().ctypes.data, nbytes, 1) != 0:
                        return 1
                return 0
            except Exception:                                     # noqa: BLE001
                return 3

        self._keep = [MG_BCAST_FN(bcast), MG_ALLGATHER_FN(allgather), MG_SEND_FN(send), MG_RECV_FN(recv), MG_GROUP_FN(group_start), MG_GROUP_FN(group_end)]
        self.ctx._chk(self.lib.gpk_mg_set_comm(self.h, None, self._keep[0], self._keep[1]))
        self.ctx._chk(self.lib.gpk_mg_set_p2p(self.h, self._keep[2], self._keep[3], self._keep[4], self._keep[5]))
        self.comm_kind = 'host-staged stand-ins over torch.distributed'

    # ---- options / calls ----------------------------------------------------------------------------------------------
    def set_option(self, key, value):
        keys = {'lookahead': 0, 'shard_hb': 1, 'col_align': 2, 'overlap_s': 3, 'p2p': 4}
        self.ctx._chk(self.lib.gpk_mg_set_option(self.h, keys[key] if isinstance(key, str) else int(key), int(value)))

    def selftest(self):
        """round trip through the bound collectives on small buffers (every rank calls it) -> True / False"""
        ok = C.c_int()
        self.ctx._chk(self.lib.gpk_mg_selftest(self.h, C.byref(ok)))
        return bool(ok.value)

    def preflight(self, nbytes=139 * 2 ** 20, reps=2):
        """Bandwidth of the BOUND collectives on a buffer of the size of one panel broadcast of BASELINE config 5 (34000 x 512 doubles =
        139 MB): every rank calls it.  -> dict with GB/s per broadcast root, of the all-gather (bytes received per rank / time) and the
        number of distinct ranks the all-gather delivered (gpk_mg_preflight)."""
        bms = (C.c_double * self.world)()
        ams, seen = C.c_double(), C.c_int()
        self.ctx._chk(self.lib.gpk_mg_preflight(self.h, int(nbytes), int(reps), bms, C.byref(ams), C.byref(seen)))
        gbs = lambda b, ms: (b / (ms * 1e-3) / 1e9) if ms > 0 else None
        per = max(nbytes // 8 // self.world, 1) * 8
        out = {'bytes': int(nbytes), 'reps': int(reps), 'bcast_ms_by_root': list(bms), 'bcast_gbs_by_root': [gbs(nbytes, m) for m in bms],
               'allgather_ms': ams.value, 'allgather_gbs_received': gbs(per * (self.world - 1), ams.value) if self.world > 1 else None,
               'ranks_seen_by_rccl': seen.value, 'p2p_bound': self.has_p2p()}
        if out['p2p_bound'] and self.world > 1:
            # the direct exchange (grouped ncclSend / ncclRecv): the same bytes per rank as the all-gather above, one transfer per peer
            pms = C.c_double()
            self.ctx._chk(self.lib.gpk_mg_preflight_p2p(self.h, int(nbytes), int(reps), C.byref(pms)))
            out['direct_ms'] = pms.value
            out['direct_gbs_received'] = gbs(per * (self.world - 1), pms.value)
        return out

    def has_p2p(self):
        return bool(self.lib.gpk_mg_has_p2p(self.h))

    def potrf(self, A_ptr, n, lda):
        """In-place lower Cholesky over all ranks (A replicated on entry, the full factor on every rank on return) -> info"""
        info = C.c_int()
        self.ctx._chk(self.lib.gpk_mg_potrf(self.h, A_ptr, int(n), int(lda), C.byref(info)))
        return self.ctx._chk_info(info.value)

    def gn_step(self, prob_struct, z_ptr, step_size, S_ptr, lds, S2_ptr, Hb_ptr, ldh, delta_ptr):
        loss, info = C.c_double(), C.c_int()
        self.ctx._chk(self.lib.gpk_mg_gn_step(self.h, C.byref(prob_struct), z_ptr, float(step_size), S_ptr, int(lds), S2_ptr, Hb_ptr, int(ldh),
                                              delta_ptr, C.byref(loss), C.byref(info)))
        return loss.value, self.ctx._chk_info(info.value)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.gpk_mg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

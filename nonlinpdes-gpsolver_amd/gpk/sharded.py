"""Multi-GPU (one process per GPU) schedule of the factor/solve path for the >=10k-point configuration, driven from Python
over torch.distributed (INTEGRATION.md route A; the same schedule natively behind the C ABI: gpk/mg.py, include/gpk_mg.h).

Only the SCHEDULE is interpreted here; every flop runs in libgpk.so through the `ops` object, and every exchange is a
torch.distributed collective on the process group it is given (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the
CPU tests, which inject a numpy `ops`).

Cholesky (Theta, and optionally the bordered Gauss-Newton matrix Hb): 1-D block-cyclic BLOCK-COLUMN distribution, panel width
nb.  The schedule is NOT written down here: it is the plan libgpk returns (gpk_mg_plan_potrf, a pure host function; the
native executor gpk_mg_potrf runs the very same list) -- a flat list of operations on three streams with explicit events:
    FACTOR k   the owner (k mod P) factors its tall panel in place          (gpk_potrf_panel_at, fused panel kernels)
    PACK / BCAST / UNPACK k   contiguous image of the panel, ncclBroadcast, copy into every other rank's matrix
                              (all ranks end up with the full L: HBM is 288 GB, Theta is 9.2 GB at config 5)
    UPDATE j,k  the owner of block column j applies panel k to it            (gpk_gemm, MFMA)
    RECORD / WAIT e           dependencies between the streams
With look-ahead (default for P > 1) the owner of panel k+1 applies panel k to that column and factors it on a high-priority
stream as soon as panel k has arrived, and the broadcast of k+1 travels on a communication stream while every rank still
applies panel k to its other columns.  No host synchronisation per panel: the pivot status stays on the device
(gpk_info_reset / gpk_info_read) and is read ONCE at the end.
Gauss-Newton step:  S = L^{-1}[A | F] is independent per right-hand-side column -> rank r solves its column range with
    the replicated L (no communication), the column shards are all-gathered through persistent staging buffers (every rank
    needs all of S for its rows of Hb = S^T S), block rows of Hb are computed cyclically and all-gathered, Hb is factored
    replicated (P <= 2: cheaper than any exchange) or with the plan above (P >= 4), and the small triangular solve + update
    are replicated, so all ranks hold the same iterate bit for bit.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a panel broadcast moves <= 139 MB at C5, the S all-gather 4.35 GB
in total, Hb 2 GB; nothing here is a ring all-reduce.
"""
import contextlib
import ctypes as C

import torch
import torch.distributed as dist

from . import mg as _mg


class Comm:
    """Thin wrapper so the schedule does not care whether a process group exists (world size 1)."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1

    def _staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == 'gloo'

    def broadcast(self, t, src):
        if self.world > 1:
            if self._staged(t):                                    # tests: several ranks on ONE GPU, gloo moves host memory
                torch.cuda.current_stream().synchronize()
            dist.broadcast(t, src=src, group=self.group)

    def all_gather_into(self, out, t):
        """out: (world * t.numel(),) contiguous; t: contiguous"""
        if self.world == 1:
            out.copy_(t.reshape(-1))
        elif self._staged(t):
            # gloo has no all_gather for device tensors (it is only used for tests: several ranks sharing ONE GPU);
            # emulate it with one broadcast per rank
            n = t.numel()
            torch.cuda.current_stream().synchronize()
            for r in range(self.world):
                part = out[r * n:(r + 1) * n]
                if r == self.rank:
                    part.copy_(t.reshape(-1))
                    torch.cuda.current_stream().synchronize()
                dist.broadcast(part, src=r, group=self.group)
        else:
            dist.all_gather_into_tensor(out, t.reshape(-1), group=self.group)

    def exchange_direct(self, parts, mine):
        """Exact-size all-to-all of one buffer per rank (the direct exchange of the sharded step, gpk_mg.hip exchange_direct): `mine`
        (contiguous) goes to every peer, parts[r] (contiguous, exactly rank r's size) receives rank r's buffer; parts[self.rank] is not
        touched.  One batch of point-to-point operations -- over RCCL a single ncclGroupStart / ncclGroupEnd, on xGMI one link per peer."""
        if self.world == 1:
            return
        staged = self._staged(mine)
        if staged:
            torch.cuda.current_stream().synchronize()
        ops = []
        for d in range(1, self.world):                             # peers in distance order, as the native executor issues them
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
            gto = dist.get_global_rank(self.group, to) if self.group is not None else to
            gfrm = dist.get_global_rank(self.group, frm) if self.group is not None else frm
            if mine.numel() > 0:
                ops.append(dist.P2POp(dist.isend, mine, gto, group=self.group))
            if parts[frm].numel() > 0:
                ops.append(dist.P2POp(dist.irecv, parts[frm], gfrm, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def first_failure(self, info, device):
        """LAPACK info over all ranks: the smallest positive index (every rank saw only the panels it factored); a negative
        value (device-side wait expired) anywhere wins."""
        if self.world == 1:
            return int(info)
        big = 1 << 40
        key = -big if info < 0 else (int(info) if info > 0 else big)
        t = torch.tensor([key], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        v = int(t.item())
        return -1 if v == -big else (0 if v == big else v)

    def max_int(self, v, device):
        if self.world == 1:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def max_float(self, v, device):
        if self.world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)


class GpuBlockOps:
    """Block operations on sub-matrices of row-major float64 torch CUDA tensors, executed by libgpk on the tensors' own
    memory (raw pointers; the library's handle runs on torch's current stream so collectives and kernels stay ordered)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.lib = ctx.lib
        self.h = ctx.h
        # one explicit (non-default) torch stream carries everything: libgpk kernels, torch copies and, through
        # torch.distributed's stream dependencies, the RCCL collectives.  (The legacy default stream has handle 0,
        # which gpk_set_stream reads as "use the handle's own stream" -- that would race with torch.)
        self.stream = torch.cuda.Stream()
        torch.cuda.set_stream(self.stream)
        ctx._chk(self.lib.gpk_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)))
        self._side = None

    # ---- streams and events of the look-ahead schedule --------------------------------------------------------------
    def streams(self, lookahead):
        """(main, panel, communication): the two side streams are high-priority torch streams, created once"""
        if not lookahead:
            return (self.stream, self.stream, self.stream)
        if self._side is None:
            self._side = (torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1))
        for s in self._side:                                       # what is in the matrix was produced on the main stream
            s.wait_stream(self.stream)
        return (self.stream, self._side[0], self._side[1])

    @contextlib.contextmanager
    def on(self, stream):
        """run libgpk and torch on `stream` inside the block"""
        prev = torch.cuda.current_stream()
        if stream == prev:
            yield
            return
        torch.cuda.set_stream(stream)
        self.ctx._chk(self.lib.gpk_set_stream(self.h, C.c_void_p(stream.cuda_stream)))
        try:
            yield
        finally:
            torch.cuda.set_stream(prev)
            self.ctx._chk(self.lib.gpk_set_stream(self.h, C.c_void_p(prev.cuda_stream)))

    def record(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def wait(self, ev):
        torch.cuda.current_stream().wait_event(ev)

    def info_reset(self):
        self.ctx._chk(self.lib.gpk_info_reset(self.h))

    def info_read(self):
        info = C.c_int()
        self.ctx._chk(self.lib.gpk_info_read(self.h, C.byref(info)))
        return info.value

    @staticmethod
    def _p(T, r=0, c=0):
        return T.data_ptr() + (r * T.stride(0) + c) * 8

    def potrf(self, A, r0, n):
        info = C.c_int()
        self.ctx._chk(self.lib.gpk_potrf(self.h, self._p(A, r0, r0), n, A.stride(0), C.byref(info)))
        return self.ctx._chk_info(info.value)

    def potrf_panel(self, A, r0, n, nrows):
        """A[r0:r0+nrows, r0:r0+n]: factor the diagonal block and solve the rows below against it (one fused panel step); the
        pivot status goes to the device-side info word (index relative to the whole matrix), no host synchronisation"""
        self.ctx._chk(self.lib.gpk_potrf_panel_at(self.h, self._p(A, r0, r0), nrows, n, A.stride(0), int(r0)))

    def trsm_right(self, A, r0, n, row0, m):
        """A[row0:row0+m, r0:r0+n] <- A[...] * L^{-T} with L = A[r0:r0+n, r0:r0+n]"""
        self.ctx._chk(self.lib.gpk_trsm_right_lt(self.h, self._p(A, r0, r0), n, A.stride(0), self._p(A, row0, r0), m, A.stride(0)))

    def update_nt(self, Cm, cr, cc, m, n, k, A, ar, ac, B, br, bc):
        """C[cr:cr+m, cc:cc+n] -= A[ar:ar+m, ac:ac+k] * B[br:br+n, bc:bc+k]^T"""
        self.ctx._chk(self.lib.gpk_gemm(self.h, 0, 1, m, n, k, -1.0, self._p(A, ar, ac), A.stride(0), self._p(B, br, bc), B.stride(0),
                                        1.0, self._p(Cm, cr, cc), Cm.stride(0)))

    def gram_tn(self, Cm, cr, cc, m, n, k, A, ac, B, bc):
        """C[cr:cr+m, cc:cc+n] = A[:k, ac:ac+m]^T * B[:k, bc:bc+n]"""
        self.ctx._chk(self.lib.gpk_gemm(self.h, 1, 0, m, n, k, 1.0, self._p(A, 0, ac), A.stride(0), self._p(B, 0, bc), B.stride(0),
                                        0.0, self._p(Cm, cr, cc), Cm.stride(0)))

    def trsm_left(self, L, n, B, c0, ncols, trans=False):
        self.ctx._chk(self.lib.gpk_trsm(self.h, int(trans), self._p(L), n, L.stride(0), self._p(B, 0, c0), ncols, B.stride(0)))

    def trsm_left_lz(self, L, n, B, c0, ncols, lead):
        """forward solve on the columns [c0, c0+ncols) of B; column c < lead (global index) is zero above row lead-1-c"""
        self.ctx._chk(self.lib.gpk_trsm_lz(self.h, self._p(L), n, L.stride(0), self._p(B, 0, c0), ncols, B.stride(0), int(lead - c0)))

    def trtri_diag(self, L, n, block=1024):
        """inverses of the block x block diagonal blocks of the factor L[:n, :n] -> (n, block) tensor (gpk_trtri_diag)"""
        D = torch.empty((n, block), dtype=torch.float64, device=L.device)
        self.ctx._chk(self.lib.gpk_trtri_diag(self.h, self._p(L), n, L.stride(0), D.data_ptr(), int(block)))
        return D

    def trsm_left_dinv(self, L, Dinv, n, B, X, c0, ncols, lead):
        """X[:, c0:c0+ncols] <- L^{-1} B[:, c0:c0+ncols] with GEMMs only (gpk_trsm_dinv); B's columns become scratch.  lead > 0:
        leading-zero right-hand sides as trsm_left_lz (lead is the global column index); X must be zero where never written."""
        self.ctx._chk(self.lib.gpk_trsm_dinv(self.h, self._p(L), Dinv.data_ptr(), Dinv.stride(0), n, L.stride(0), self._p(B, 0, c0),
                                             ncols, B.stride(0), self._p(X, 0, c0), X.stride(0), int(max(lead - c0, 0)) if lead else 0))

    def gram_tn_lz(self, Cm, cr, cc, m, n, k, A, ac, B, bc, lead):
        """gram_tn with operand B's column c < lead (global index) zero above row lead-1-c"""
        self.ctx._chk(self.lib.gpk_gemm_lz(self.h, 1, m, n, k, 1.0, self._p(A, 0, ac), A.stride(0), self._p(B, 0, bc), B.stride(0),
                                           0.0, self._p(Cm, cr, cc), Cm.stride(0), int(lead - bc)))

    def trsv(self, L, n, x, trans):
        self.ctx._chk(self.lib.gpk_trsm(self.h, int(trans), self._p(L), n, L.stride(0), x.data_ptr(), 1, 1))

    def gn_build(self, prob_struct, z, S, rev=False):
        fn = self.lib.gpk_gn_build_rev if rev else self.lib.gpk_gn_build
        self.ctx._chk(fn(self.h, C.byref(prob_struct), z.data_ptr(), self._p(S), S.stride(0)))

    def axpy(self, n, alpha, x, y):
        self.ctx._chk(self.lib.gpk_axpy(self.h, n, float(alpha), x.data_ptr(), y.data_ptr()))


def _ceil_div(a, b):
    return (a + b - 1) // b


class ShardedFactorSolve:
    def __init__(self, ops, comm, nb=512, lookahead=None, shard_hb=None, direct=False):
        """lookahead: None = on for more than one rank; shard_hb: None = the Cholesky of Hb is panel-sharded from 4 ranks on
        (replicated below: every rank factors its own copy, no communication); direct: the two exchanges of the step (column shards of S,
        block rows of Hb) as exact-size point-to-point batches instead of padded all-gathers (round 6; gpk_mg option key 3 = 2)."""
        self.direct = bool(direct)
        self.ops, self.comm, self.nb = ops, comm, int(nb)
        self.rank, self.P = comm.rank, comm.world
        self.lookahead = (self.P > 1) if lookahead is None else bool(lookahead)
        self.shard_hb = (self.P >= 4) if shard_hb is None else bool(shard_hb)
        self.col_align = 128                                       # column shards start at multiples of this (tile/vector alignment)
        self._panel = [None, None]                                 # transfer buffers of the panel broadcasts (slot = panel mod 2)
        self._stage = {}                                           # persistent all-gather staging buffers, by name
        self.trace = None                                          # tests: set to a list to record the operations as issued

    def describe(self, world, dinv_block):
        hb = 'panel-sharded POTRF(Hb) with the same plan' if self.shard_hb else 'replicated POTRF(Hb)'
        la = ('with look-ahead (next panel factored and broadcast on side streams while the trailing updates run)'
              if self.lookahead else 'no look-ahead')
        return (f'Theta: panel-sharded Cholesky (block-cyclic columns, width {self.nb}, RCCL broadcast, {la}); step: column-sharded TRSM '
                f'(GEMM-only, inverted {dinv_block}-row diagonal blocks of the factor) + all-gather(S) + row-block-sharded SYRK + '
                f'all-gather(Hb) + {hb} + replicated TRSV over {world} rank(s)')

    # ------------------------------------------------------------------------------------------------ buffers
    def _panel_buf(self, slot, like, count):
        b = self._panel[slot]
        if b is None or b.numel() < count or b.device != like.device:
            b = torch.empty(count, dtype=torch.float64, device=like.device)
            self._panel[slot] = b
        return b

    def _staging(self, name, like, count):
        b = self._stage.get(name)
        if b is None or b.numel() < count or b.device != like.device:
            b = torch.empty(count, dtype=torch.float64, device=like.device)
            self._stage[name] = b
        return b[:count]

    # ------------------------------------------------------------------------------------------------ Cholesky
    def potrf(self, A, n):
        """In-place lower Cholesky of A[:n, :n] (row-major torch tensor, any leading dimension), panels owned
        block-cyclically; on return every rank holds the complete factor.  Returns LAPACK-style info (identical on all ranks)."""
        nb, P, rank, ops = self.nb, self.P, self.rank, self.ops
        nblk = _ceil_div(n, nb)
        la = self.lookahead and nblk > 1
        plan = _mg.plan_potrf(n, nb, P, rank, la)
        have_streams = hasattr(ops, 'streams')
        # the device-side pivot status is cleared on the main stream BEFORE the side streams are made to wait on it: FACTOR 0 runs on
        # the panel stream with no WAIT in the plan, and must be ordered after the memset (the native executor does the same)
        if hasattr(ops, 'info_reset'):
            ops.info_reset()
        streams = ops.streams(la) if have_streams else (None, None, None)
        on = ops.on if have_streams else (lambda s: contextlib.nullcontext())
        events = {}
        cap = n * min(nb, n)                                       # panel 0 is the largest
        for kind, a, b, s in plan:
            if self.trace is not None:
                self.trace.append((_mg.OP_NAMES[kind], a, b, s))
            k = b if kind == _mg.UPDATE else a
            k0 = k * nb
            kb = min(nb, n - k0)
            with on(streams[s]):
                if kind == _mg.FACTOR:
                    ops.potrf_panel(A, k0, kb, n - k0)             # diagonal block + the rows below, fused panel kernels
                elif kind == _mg.PACK:
                    self._panel_buf(k & 1, A, cap)[:(n - k0) * kb].view(n - k0, kb).copy_(A[k0:n, k0:k0 + kb])
                elif kind == _mg.BCAST:
                    self.comm.broadcast(self._panel_buf(k & 1, A, cap)[:(n - k0) * kb], b)
                elif kind == _mg.UNPACK:
                    A[k0:n, k0:k0 + kb].copy_(self._panel_buf(k & 1, A, cap)[:(n - k0) * kb].view(n - k0, kb))
                elif kind == _mg.UPDATE:
                    j0 = a * nb
                    jb = min(nb, n - j0)
                    ops.update_nt(A, j0, j0, n - j0, jb, kb, A, j0, k0, A, j0, k0)
                elif kind == _mg.RECORD:
                    events[a] = ops.record() if have_streams else True
                elif kind == _mg.WAIT:
                    if have_streams:
                        ops.wait(events[a])
                    else:
                        assert events.get(a), 'plan waits for an event that was never recorded'
        info = ops.info_read() if hasattr(ops, 'info_read') else 0     # ONE host read for the whole factorisation
        return self.comm.first_failure(info, A.device)

    # ------------------------------------------------------------------------------------------------ GN step
    def column_range(self, ncols):
        """contiguous column shard of the right-hand sides, multiples of 128 columns except the last"""
        per = _ceil_div(_ceil_div(ncols, self.P), 128) * 128
        c0 = min(self.rank * per, ncols)
        return c0, min(c0 + per, ncols), per

    def column_ranges_lz(self, ncols, lead, rows):
        """Contiguous column shards of equal WORK for the leading-zero right-hand side: column c < lead starts at row
        lead-1-c, so its forward solve costs ~(rows - start)^2.  Returns the P+1 boundaries (multiples of col_align);
        computed by the library (gpk_mg_column_bounds: the native step uses the same function)."""
        return _mg.column_bounds(ncols, lead, rows, self.P, self.col_align)

    def gn_step(self, prob_struct, nz, rows, L, z, S, Hb, delta, step_size, rev=False, Dinv=None, S2=None):
        """One Gauss-Newton step with S column-sharded and Hb row-block-sharded; z is updated identically on all ranks.
        L: replicated factor (rows x rows); S: (rows, >= nz+1); Hb: (nz+1, >= nz+1); returns (loss_in, info).
        rev (elliptic system): unknown j lives in column nz-1-j of S, which makes column c zero above row nz-1-c; the
        solves and products skip those zeros (28 % of the flops) and the column shards are cut by work, not by width.
        Dinv (ops.trtri_diag(L)) + S2 (zero-initialised, same shape as S, reused across steps): the column solves run through
        the inverted diagonal blocks (GEMMs only) out of place into S2, which then takes S's role; S is scratch."""
        ops, comm, P, rank, nb = self.ops, self.comm, self.P, self.rank, self.nb
        nc = nz + 1
        if rev:
            ops.gn_build(prob_struct, z, S, rev=True)              # every rank writes all of [A | F] (a memset + O(N))
            bounds = self.column_ranges_lz(nc, nz, rows)
        else:
            ops.gn_build(prob_struct, z, S)
            per0 = _ceil_div(_ceil_div(nc, P), self.col_align) * self.col_align
            bounds = [min(r * per0, nc) for r in range(P)] + [nc]
        c0, c1 = bounds[rank], bounds[rank + 1]
        per = max(bounds[r + 1] - bounds[r] for r in range(P))
        use_dinv = Dinv is not None and S2 is not None
        if c1 > c0:                                                # my columns of L^{-1}[A | F]
            if use_dinv:
                ops.trsm_left_dinv(L, Dinv, rows, S, S2, c0, c1 - c0, nz if rev else 0)
            elif rev:
                ops.trsm_left_lz(L, rows, S, c0, c1 - c0, nz)
            else:
                ops.trsm_left(L, rows, S, c0, c1 - c0)
        if use_dinv:
            S = S2                                                 # the solved block lives in S2 from here on
        if P > 1 and self.direct:                                  # exact sizes, one transfer per peer (no padding to the widest shard)
            widths = [bounds[r + 1] - bounds[r] for r in range(P)]
            mine = self._staging('s_send', S, rows * widths[rank]).view(rows, widths[rank])
            if c1 > c0:
                mine.copy_(S[:, c0:c1])
            allp = self._staging('s_recv', S, rows * nc)
            offs = [rows * bounds[r] for r in range(P)]
            parts = [allp[offs[r]:offs[r] + rows * widths[r]] for r in range(P)]
            comm.exchange_direct(parts, mine.reshape(-1))
            for r in range(P):
                if r != rank and widths[r] > 0:
                    S[:, bounds[r]:bounds[r + 1]].copy_(parts[r].view(rows, widths[r]))
        elif P > 1:                                                # all-gather the column shards of S (padded to the widest)
            mine = self._staging('s_send', S, rows * per).view(rows, per)
            if c1 > c0:
                mine[:, :c1 - c0].copy_(S[:, c0:c1])
            allp = self._staging('s_recv', S, P * rows * per)
            comm.all_gather_into(allp, mine)
            for r in range(P):
                a, b = bounds[r], bounds[r + 1]
                if r != rank and b > a:
                    S[:, a:b].copy_(allp[r * rows * per:(r + 1) * rows * per].view(rows, per)[:, :b - a])
        # Hb = S^T S, lower block rows i (cyclic over ranks): Hb[i-block, 0:(i+1)nb]
        nblk = _ceil_div(nc, nb)
        for i in range(nblk):
            if i % P != rank:
                continue
            i0 = i * nb
            ib = min(nb, nc - i0)
            if rev:
                ops.gram_tn_lz(Hb, i0, 0, ib, i0 + ib, rows, S, i0, S, 0, nz)
            else:
                ops.gram_tn(Hb, i0, 0, ib, i0 + ib, rows, S, i0, S, 0)
        if P > 1:
            # all-gather of the block rows, LOWER parts only (round 4; whole rows until round 3: twice the bytes): block row i travels as
            # ib x (i0 + ib) packed entries, every rank's share padded to the largest -- the layout of the native step (gpk_mg.hip)
            blocks = [(i * nb, min(nb, nc - i * nb)) for i in range(nblk)]
            share = [sum(ib * (i0 + ib) for i, (i0, ib) in enumerate(blocks) if i % P == r) for r in range(P)]
            hshare = max(share)
            mine = self._staging('h_send', Hb, hshare)
            o = 0
            for i in range(rank, nblk, P):
                i0, ib = blocks[i]
                mine[o:o + ib * (i0 + ib)].view(ib, i0 + ib).copy_(Hb[i0:i0 + ib, :i0 + ib])
                o += ib * (i0 + ib)
            allp = self._staging('h_recv', Hb, P * hshare)
            if self.direct:                                        # exact shares, packed back to back
                hoff = [sum(share[:r]) for r in range(P)]
                comm.exchange_direct([allp[hoff[r]:hoff[r] + share[r]] for r in range(P)], mine[:share[rank]])
            else:
                hoff = [r * hshare for r in range(P)]
                comm.all_gather_into(allp, mine)
            for r in range(P):
                if r == rank:
                    continue
                o = hoff[r]
                for i in range(r, nblk, P):
                    i0, ib = blocks[i]
                    Hb[i0:i0 + ib, :i0 + ib].copy_(allp[o:o + ib * (i0 + ib)].view(ib, i0 + ib))
                    o += ib * (i0 + ib)
        loss_dev = Hb[nz, nz].clone()                              # read on the host at the END of the step (one synchronisation)
        # Cholesky of the bordered matrix.  Replicated (every rank factors its own copy, no communication): at n_z = 16000 one GPU
        # needs ~37 ms, while the panel scheme pays per 512-wide panel an owner-only factorisation and a broadcast on top of
        # update/P -- with look-ahead those hide behind the updates, which is why from 4 ranks on the plan of the big
        # factorisation is used for Hb as well (shard_hb).  The last row of the factor is (L_H^{-1} g / 2)^T either way.
        if self.shard_hb and P > 1:
            info = self.potrf(Hb, nc)
        else:
            info = comm.max_int(ops.potrf(Hb, 0, nc), Hb.device)
        if info == nc:
            info = 0                                               # the border pivot is not part of H
        delta.copy_(Hb[nz, :nz])
        ops.trsv(Hb, nz, delta, True)                              # replicated: L_H^{-T} y
        if rev:
            delta.copy_(torch.flip(delta, dims=[0]))               # back to the natural order of the unknowns
        ops.axpy(nz, -float(step_size), delta, z)
        return float(loss_dev.item()), info

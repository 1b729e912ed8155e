"""Multi-GPU (one process per GPU) schedule of the factor/solve path for the >=10k-point configuration.

Only the SCHEDULE lives here (which rank factors which panel, what is broadcast, who updates which block column); every
flop runs in libgpk.so through the `ops` object, and every exchange is a torch.distributed collective on the process
group it is given (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests, which inject a numpy `ops`).

Cholesky (Theta, and the bordered Gauss-Newton matrix Hb):  1-D block-cyclic BLOCK-COLUMN distribution, panel width nb.
    for k in panels:  owner = k mod P
        owner:   potrf(diagonal block), rows below <- rows * L_kk^{-T}          (gpk_potrf, gpk_trsm_right_lt)
        all:     broadcast the factored panel ((n - k nb) x nb) from its owner   (ncclBroadcast)
        all:     store the panel (every rank ends up with the full L: HBM is 288 GB, Theta is 9.2 GB at C5)
        rank r:  update the block columns j > k with j mod P == r                (gpk_gemm, MFMA)
Gauss-Newton step:  S = L^{-1}[A | F] is independent per right-hand-side column -> rank r solves its column range with
    the replicated L (no communication), the column shards are all-gathered (every rank needs all of S for its rows of
    Hb = S^T S), block rows of Hb are computed cyclically and all-gathered, Hb is factored REPLICATED on every rank (cheaper
    than the panel scheme at this size, see gn_step) and the small triangular solve + update are replicated, so all ranks
    hold the same iterate bit for bit.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a panel broadcast moves <= 139 MB at C5, the S all-gather 4.35 GB
in total, Hb 2 GB; nothing here is a ring all-reduce.
"""
import ctypes as C

import torch
import torch.distributed as dist


class Comm:
    """Thin wrapper so the schedule does not care whether a process group exists (world size 1)."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1

    def broadcast(self, t, src):
        if self.world > 1:
            dist.broadcast(t, src=src, group=self.group)

    def all_gather(self, outs, t):
        if self.world == 1:
            outs[0].copy_(t)
        elif t.is_cuda and dist.get_backend(self.group) == 'gloo':
            # gloo has no all_gather for device tensors (it is only used for tests: several ranks sharing ONE GPU);
            # emulate it with one broadcast per rank
            for r in range(self.world):
                if r == self.rank:
                    outs[r].copy_(t)
                dist.broadcast(outs[r], src=r, group=self.group)
        else:
            dist.all_gather(outs, t, group=self.group)

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

    @staticmethod
    def _p(T, r=0, c=0):
        return T.data_ptr() + (r * T.stride(0) + c) * 8

    def potrf(self, A, r0, n):
        info = C.c_int()
        self.ctx._chk(self.lib.gpk_potrf(self.h, self._p(A, r0, r0), n, A.stride(0), C.byref(info)))
        return self.ctx._chk_info(info.value)

    def potrf_panel(self, A, r0, n, nrows):
        """A[r0:r0+nrows, r0:r0+n]: factor the diagonal block and solve the rows below against it (one fused panel step)"""
        info = C.c_int()
        self.ctx._chk(self.lib.gpk_potrf_panel(self.h, self._p(A, r0, r0), nrows, n, A.stride(0), C.byref(info)))
        return self.ctx._chk_info(info.value)

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
    def __init__(self, ops, comm, nb=512):
        self.ops, self.comm, self.nb = ops, comm, int(nb)
        self.rank, self.P = comm.rank, comm.world
        self.col_align = 128                                       # column shards start at multiples of this (tile/vector alignment)
        self._panel = None

    def describe(self, world, dinv_block):
        return (f'Theta: panel-sharded Cholesky (block-cyclic columns, width {self.nb}, RCCL broadcast); step: column-sharded TRSM '
                f'(GEMM-only, inverted {dinv_block}-row diagonal blocks of the factor) + all-gather(S) + row-block-sharded SYRK + '
                f'all-gather(Hb) + replicated POTRF(Hb)/TRSV over {world} rank(s)')

    # ------------------------------------------------------------------------------------------------ Cholesky
    def _panel_buf(self, A, rows, cols):
        need = rows * cols
        if self._panel is None or self._panel.numel() < need or self._panel.device != A.device:
            self._panel = torch.empty(need, dtype=torch.float64, device=A.device)
        return self._panel[:need].view(rows, cols)

    def potrf(self, A, n):
        """In-place lower Cholesky of A[:n, :n] (row-major torch tensor, any leading dimension), panels owned
        block-cyclically; on return every rank holds the complete factor.  Returns LAPACK-style info (max over ranks)."""
        nb, P, rank, ops = self.nb, self.P, self.rank, self.ops
        nblk = _ceil_div(n, nb)
        info = 0
        for k in range(nblk):
            k0 = k * nb
            kb = min(nb, n - k0)
            below = n - (k0 + kb)
            owner = k % P
            panel = self._panel_buf(A, n - k0, kb)
            if rank == owner:
                i = ops.potrf_panel(A, k0, kb, n - k0)             # diagonal block + the rows below, fused panel kernels
                if i and not info:
                    info = k0 + i
                if P > 1:
                    panel.copy_(A[k0:n, k0:k0 + kb])
            if P > 1:
                self.comm.broadcast(panel, owner)
                if rank != owner:
                    A[k0:n, k0:k0 + kb].copy_(panel)
            for j in range(k + 1, nblk):                       # right-looking update of MY block columns
                if j % P != rank:
                    continue
                j0 = j * nb
                jb = min(nb, n - j0)
                ops.update_nt(A, j0, j0, n - j0, jb, kb, A, j0, k0, A, j0, k0)
        return self.comm.max_int(info, A.device)

    # ------------------------------------------------------------------------------------------------ GN step
    def column_range(self, ncols):
        """contiguous column shard of the right-hand sides, multiples of 128 columns except the last"""
        per = _ceil_div(_ceil_div(ncols, self.P), 128) * 128
        c0 = min(self.rank * per, ncols)
        return c0, min(c0 + per, ncols), per

    def column_ranges_lz(self, ncols, lead, rows):
        """Contiguous column shards of equal WORK for the leading-zero right-hand side: column c < lead starts at row
        lead-1-c, so its forward solve costs ~(rows - start)^2.  Returns the P+1 boundaries (multiples of 128)."""
        import numpy as np
        c = np.arange(ncols)
        start = np.maximum(0, lead - 1 - c)
        w = np.cumsum((rows - start).astype(np.float64) ** 2)
        bounds = [0]
        for r in range(1, self.P):
            cut = int(np.searchsorted(w, w[-1] * r / self.P))
            cut = min(max(_ceil_div(cut, self.col_align) * self.col_align, bounds[-1]), ncols)
            bounds.append(cut)
        bounds.append(ncols)
        return bounds

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
        if P > 1:                                                  # all-gather the column shards of S (padded to the widest)
            mine = torch.zeros((rows, per), dtype=torch.float64, device=S.device)
            if c1 > c0:
                mine[:, :c1 - c0].copy_(S[:, c0:c1])
            parts = [torch.empty_like(mine) for _ in range(P)]
            comm.all_gather(parts, mine)
            for r in range(P):
                a, b = bounds[r], bounds[r + 1]
                if r != rank and b > a:
                    S[:, a:b].copy_(parts[r][:, :b - a])
            del parts, mine
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
        if P > 1:                                                  # all-gather the block rows (padded to equal counts)
            per_rank = _ceil_div(nblk, P)
            width = Hb.stride(0)
            mine = torch.zeros((per_rank * nb, width), dtype=torch.float64, device=Hb.device)
            for t, i in enumerate(range(rank, nblk, P)):
                i0 = i * nb; ib = min(nb, nc - i0)
                mine[t * nb:t * nb + ib, :nc].copy_(Hb[i0:i0 + ib, :nc])
            parts = [torch.empty_like(mine) for _ in range(P)]
            comm.all_gather(parts, mine)
            for r in range(P):
                if r == rank:
                    continue
                for t, i in enumerate(range(r, nblk, P)):
                    i0 = i * nb; ib = min(nb, nc - i0)
                    Hb[i0:i0 + ib, :nc].copy_(parts[r][t * nb:t * nb + ib, :nc])
            del parts, mine
        loss_in = float(Hb[nz, nz].item())
        # Every rank now holds all of Hb and factors it locally (replicated, no communication): at n_z = 16000 one GPU
        # needs ~37 ms (27 ms of GEMM + the 250-panel latency chain), while the panel-broadcast scheme pays per 512-wide
        # panel an owner-only factorisation (~0.9 ms) and a broadcast (~0.4 ms) on top of update/P -- 32 panels cost
        # more than the replicated factorisation at every P >= 2.  (Theta itself is factored once with the sharded scheme.)
        info = ops.potrf(Hb, 0, nc)                                # bordered: last row of the factor = L_H^{-1} g / 2
        if info == nc:
            info = 0                                               # the border pivot is not part of H
        info = comm.max_int(info, Hb.device)
        delta.copy_(Hb[nz, :nz])
        ops.trsv(Hb, nz, delta, True)                              # replicated: L_H^{-T} y
        if rev:
            delta.copy_(torch.flip(delta, dims=[0]))               # back to the natural order of the unknowns
        ops.axpy(nz, -float(step_size), delta, z)
        return loss_in, info

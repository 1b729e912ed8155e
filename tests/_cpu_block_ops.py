"""TEST DOUBLE for gpk.sharded.GpuBlockOps: the same block-operation interface on CPU torch tensors with numpy/scipy,
so that the multi-process SCHEDULE of gpk/sharded.py (panel ownership, broadcasts, all-gathers) can be exercised with the
gloo backend on a machine without GPUs.  Lives under tests/ only; the product has no CPU path."""
import numpy as np
from scipy.linalg import solve_triangular


class NumpyBlockOps:
    def __init__(self, oracle_system=None):
        self.sys = oracle_system
        self.info = 0                                         # the device-side info word of the real handle

    def info_reset(self):
        self.info = 0

    def info_read(self):
        return self.info

    def potrf(self, A, r0, n):
        a = A.numpy()
        blk = a[r0:r0 + n, r0:r0 + n]
        try:
            L = np.linalg.cholesky(np.tril(blk) + np.tril(blk, -1).T)
        except np.linalg.LinAlgError:
            blk[:] = np.nan
            return 1
        blk[np.tril_indices(n)] = L[np.tril_indices(n)]
        return 0

    def potrf_panel(self, A, r0, n, nrows):
        info = self.potrf(A, r0, n)
        if nrows > n and info == 0:
            self.trsm_right(A, r0, n, r0 + n, nrows - n)
        if info and not self.info:
            self.info = r0 + info                             # first failure wins, index relative to the whole matrix
        return info

    def trsm_right(self, A, r0, n, row0, m):
        a = A.numpy()
        L = np.tril(a[r0:r0 + n, r0:r0 + n])
        X = a[row0:row0 + m, r0:r0 + n]
        X[:] = solve_triangular(L, X.T, lower=True, check_finite=False).T

    def update_nt(self, Cm, cr, cc, m, n, k, A, ar, ac, B, br, bc):
        Cm.numpy()[cr:cr + m, cc:cc + n] -= A.numpy()[ar:ar + m, ac:ac + k] @ B.numpy()[br:br + n, bc:bc + k].T

    def gram_tn(self, Cm, cr, cc, m, n, k, A, ac, B, bc):
        Cm.numpy()[cr:cr + m, cc:cc + n] = A.numpy()[:k, ac:ac + m].T @ B.numpy()[:k, bc:bc + n]

    def trsm_left(self, L, n, B, c0, ncols, trans=False):
        b = B.numpy()
        b[:n, c0:c0 + ncols] = solve_triangular(np.tril(L.numpy()[:n, :n]), b[:n, c0:c0 + ncols], lower=True,
                                                trans='T' if trans else 'N', check_finite=False)

    def trsv(self, L, n, x, trans):
        x.numpy()[:n] = solve_triangular(np.tril(L.numpy()[:n, :n]), x.numpy()[:n], lower=True, trans='T' if trans else 'N', check_finite=False)

    def gn_build(self, prob_struct, z, S, rev=False):
        s = S.numpy()
        s[:] = 0.0
        zz = z.numpy()
        A = self.sys.A(zz)[0]
        F = self.sys.F(zz)[0]
        s[:A.shape[0], :A.shape[1]] = A[:, ::-1] if rev else A     # rev: unknown j in column nz-1-j
        s[:F.size, A.shape[1]] = F

    # the leading-zero variants compute the same thing (the zeros are structural); the double also CHECKS the structure
    def trsm_left_lz(self, L, n, B, c0, ncols, lead):
        b = B.numpy()
        for c in range(c0, min(c0 + ncols, lead)):
            assert not np.any(b[:max(0, lead - 1 - c), c]), 'leading-zero structure violated'
        self.trsm_left(L, n, B, c0, ncols)
        for c in range(c0, min(c0 + ncols, lead)):
            assert not np.any(b[:max(0, lead - 1 - c), c])

    def trtri_diag(self, L, n, block=1024):
        import torch
        D = np.zeros((n, block))
        l = np.tril(L.numpy()[:n, :n])
        for k0 in range(0, n, block):
            nk = min(block, n - k0)
            D[k0:k0 + nk, :nk] = solve_triangular(l[k0:k0 + nk, k0:k0 + nk], np.eye(nk), lower=True, check_finite=False)
        return torch.from_numpy(D)

    def trsm_left_dinv(self, L, Dinv, n, B, X, c0, ncols, lead):
        """blocked forward substitution through the inverted diagonal blocks, out of place (B becomes scratch); checks that the
        part of X it does not write (left of the leading-zero boundary) is zero on entry"""
        b, x, l, d = B.numpy(), X.numpy(), L.numpy(), Dinv.numpy()
        db = d.shape[1]
        cols = slice(c0, c0 + ncols)
        for c in range(c0, min(c0 + ncols, lead)):
            assert not np.any(x[:max(0, lead - 1 - c), c]), 'X not zero above the leading-zero boundary'
        for k0 in range(0, n, db):
            nk = min(db, n - k0)
            x[k0:k0 + nk, cols] = d[k0:k0 + nk, :nk] @ b[k0:k0 + nk, cols]
            if k0 + nk < n:
                b[k0 + nk:n, cols] -= l[k0 + nk:n, k0:k0 + nk] @ x[k0:k0 + nk, cols]
        b[:n, cols] = np.nan                                       # scratch: nobody may read it afterwards

    def gram_tn_lz(self, Cm, cr, cc, m, n, k, A, ac, B, bc, lead):
        self.gram_tn(Cm, cr, cc, m, n, k, A, ac, B, bc)

    def axpy(self, n, alpha, x, y):
        y.numpy()[:n] += alpha * x.numpy()[:n]

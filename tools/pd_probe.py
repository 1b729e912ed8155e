#!/usr/bin/env python3
"""Is the C2-size Gram matrix (sigma 0.2, nugget 1e-13) numerically positive definite, and for whom?
Compares the HIP factorisation with LAPACK (numpy) on the SAME device-assembled matrix."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
Nd, Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4000, int(sys.argv[2]) if len(sys.argv) > 2 else 400
for seed in (0, 1):
    np.random.seed(seed)
    Xd = np.random.uniform(0, 1, (Nd, 2)); Xb = np.random.uniform(0, 1, (Nb, 2))
    for nug in (1e-13, 1e-12, 1e-11, 1e-10):
        T, r = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, nug, 'adaptive')
        A = T.download()
        info = ctx.potrf(T)
        Lg = np.tril(T.download())
        t = time.time()
        try:
            Lc = np.linalg.cholesky(A); ok = True
        except np.linalg.LinAlgError:
            ok = False
        msg = f'seed {seed} nugget {nug:g}: hip info={info} lapack_ok={ok} ({time.time()-t:.1f}s)'
        if ok:
            dc = np.diag(Lc)
            msg += f' min diag lapack {dc.min():.3e}'
            if info == 0:
                dg = np.diag(Lg)
                msg += f' hip {dg.min():.3e} max rel diag diff {np.max(np.abs(dg-dc)/dc):.2e} ||L-Lc||/||Lc|| {np.linalg.norm(Lg-Lc)/np.linalg.norm(Lc):.2e}'
                msg += f' resid hip {np.linalg.norm(Lg@Lg.T-A)/np.linalg.norm(A):.2e} lapack {np.linalg.norm(Lc@Lc.T-A)/np.linalg.norm(A):.2e}'
        print(msg, flush=True)
        T.free()

import os, sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/nonlinpdes-gpsolver_amd')
import gpk
from src.sample_points import sampled_pts_rdm
ctx = gpk.Context(0)
for Nd, Nb in ((2000,200),(4000,400),(5000,1),(10000,1000)):
    np.random.seed(0); Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0,1],[0,1]])); N=2*Nd+Nb
    T = ctx.empty(N,N)
    for f in (1,0,1,0):
        ctx.lib.gpk_debug_set(48,f); best=1e9
        for r in range(4):
            ctx.assemble('Nonlinear_elliptic','Gaussian',0.2,Xd,Xb,1e-10,'adaptive',out=T)
            ctx.timer_start(); info=ctx.potrf(T); best=min(best,ctx.timer_stop())
        print(f'N={N} fused={f}: {best:.3f} ms info {info}', flush=True)
    T.free()

#!/usr/bin/env python3
"""gpk_mg_potrf on ONE rank with and without the look-ahead plan (three streams, events) at the sizes of config 5 / the north-star
target: wall time per factorisation and, under `rocprofv3 --kernel-trace`, the evidence that nothing synchronises with the host per
panel (tools/mg_trace_summary.py reads the trace).  Usage: mg_lookahead_probe.py [N_domain N_boundary] [panel]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
from gpk.mg import MultiGpu
from src.sample_points import sampled_pts_rdm
Nd = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
Nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
panel = int(sys.argv[3]) if len(sys.argv) > 3 else 512
modes = [int(m) for m in os.environ.get('MODES', '0,1').split(',')]
ctx = gpk.Context(0)
np.random.seed(0)
Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]))
N = 2 * Nd + Nb
T = ctx.empty(N, N)
mg = MultiGpu(ctx, 0, 1, panel=panel)
for la in modes:
    mg.set_option('lookahead', la)
    for rep in range(3):
        ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-12, 'adaptive', out=T)
        ctx.synchronize(); t0 = time.perf_counter()
        info = mg.potrf(T.ptr, N, T.ld)
        ms = 1e3 * (time.perf_counter() - t0)
        print(f'N={N} panel={panel} lookahead={la} rep={rep}: {ms:.1f} ms, info {info}, {N**3/3/ms/1e9:.1f} TF/s', flush=True)
ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-12, 'adaptive', out=T)
ctx.synchronize(); t0 = time.perf_counter(); info = ctx.potrf(T); ms = 1e3 * (time.perf_counter() - t0)
print(f'N={N} gpk_potrf (single-GPU routine): {ms:.1f} ms, info {info}, {N**3/3/ms/1e9:.1f} TF/s')

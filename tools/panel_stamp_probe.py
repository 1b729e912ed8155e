#!/usr/bin/env python3
"""Phase breakdown (shader-clock stamps of workgroup 1) of the fused Cholesky panel kernel: staging round trip vs potf2_tile<TALL>."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0, dev=True)
buf = (C.c_ulonglong * 16)()
ctx.lib.gpk_debug_stamps(ctx.h, None, 1)
rng = np.random.RandomState(0)
n = 512
M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n)
for rep in range(4):
    dA = ctx.array(A); info = ctx.potrf(dA); ctx.synchronize()
    ctx.lib.gpk_debug_stamps(ctx.h, buf, 1)
    s = list(buf)
    print('panel kernel (last launch with >= 2 workgroups), workgroup 1: stage %d cycles, potf2_tile<TALL> %d cycles' % (s[1] - s[0], s[2] - s[1]))
    print('   second design, panel 2 (thread 0 = wave 0): factor %d | wait for barrier B %d | update+reload+stage %d | to barrier B of panel 3 %d' % (
        s[4] - s[3], s[5] - s[4], s[6] - s[5], s[7] - s[6]))
    print('   third design: tile wave 0 work %d | barrier wait %d ; factor wave: factor+store %d | wait %d | apply %d' % (s[4] - s[3], s[5] - s[4], s[9] - s[8], s[10] - s[9], s[11] - s[10]))
ctx.lib.gpk_debug_stamps(ctx.h, None, 0)

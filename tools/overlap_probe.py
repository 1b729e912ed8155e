#!/usr/bin/env python3
"""Does a SYRK on a low-priority side stream hide behind the latency-bound Cholesky panel chain? (development probe)"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0, dev=True)
lib = C.CDLL(os.path.join(ROOT, 'nonlinpdes-gpsolver_amd', 'csrc', 'libgpk.so'))
rng = np.random.RandomState(0)
n, k = 4001, 8400
S = ctx.array(rng.normal(size=(k, n)) / np.sqrt(k))
H = ctx.empty(n, n); C2 = ctx.empty(n, n)
ctx.syrk(n, k, 1.0, S, 0.0, H)                                  # H = S^T S (SPD, lower)
ctx.synchronize()
ms = (C.c_double * 3)()
lib.gpk_debug_overlap_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
for extra in (0, 0, 512, 16384, 45056):
    ctx.lib.gpk_debug_set(9, extra)
    rc = lib.gpk_debug_overlap_probe(ctx.h, H.ptr, n, H.ld, S.ptr, k, S.ld, C2.ptr, C2.ld, ms)
    print('extra LDS', extra, 'rc', rc, 'potrf alone %.2f ms | syrk alone %.2f ms | concurrent %.2f ms' % tuple(ms))
ctx.lib.gpk_debug_set(9, 0)
for cus in (16, 32, 64):
    ctx.lib.gpk_debug_set(11, cus)
    rc = lib.gpk_debug_overlap_probe(ctx.h, H.ptr, n, H.ld, S.ptr, k, S.ld, C2.ptr, C2.ld, ms)
    print('CU-mask partition: chain stream on', cus, 'CUs, rc', rc, 'potrf alone %.2f ms | syrk alone %.2f ms | concurrent %.2f ms' % tuple(ms))
ctx.lib.gpk_debug_set(11, 0)

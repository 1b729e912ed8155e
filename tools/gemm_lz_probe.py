#!/usr/bin/env python3
"""Dense vs leading-zero NN GEMM at the shape of the big update of the solve recursion (m x 4001 x k), sustained (20 launches back
to back): executed TFLOP/s.  Keys: 16 = walk K downwards.  Development probe."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
rng = np.random.RandomState(0)
def executed(m, n, k, lead):
    tot = 0.0
    for n0 in range(0, n, 64):
        bn = min(64, n - n0)
        k0 = (max(0, lead - (n0 + 64)) // 16) * 16 if lead > 0 else 0
        tot += 2.0 * m * bn * max(0, k - min(k0, k))
    return tot
for (m, n, k) in ((3280, 4001, 5120), (10760, 10001, 10240)):
    A = ctx.array(rng.normal(size=(m, k))); B = ctx.array(rng.normal(size=(k, n))); Cm = ctx.empty(m, n)
    for lead in (0, n - 1):
        for rev in (0, 1):
            if lead == 0 and rev: continue
            ctx.lib.gpk_debug_set(16, rev)
            ctx.lib.gpk_gemm_lz(ctx.h, 0, m, n, k, -1.0, A.ptr, A.ld, B.ptr, B.ld, 1.0, Cm.ptr, Cm.ld, lead); ctx.synchronize()
            reps = 10
            ctx.timer_start()
            for _ in range(reps):
                ctx.lib.gpk_gemm_lz(ctx.h, 0, m, n, k, -1.0, A.ptr, A.ld, B.ptr, B.ld, 1.0, Cm.ptr, Cm.ld, lead)
            ms = ctx.timer_stop() / reps
            print(f'm={m} n={n} k={k} lead={lead} rev_k={rev}: {ms:8.3f} ms  executed {executed(m, n, k, lead) / ms / 1e9:6.1f} TF/s')
    A.free(); B.free(); Cm.free()
ctx.lib.gpk_debug_set(16, 0)

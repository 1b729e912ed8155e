#!/usr/bin/env python3
"""Cholesky of an SPD matrix of order n: one stream / whole chip vs the two-partition pipeline (gpk_debug_set 20 = max order,
13 = CUs of the chain partition, 18 = left-looking updates inside a block column).  Development probe."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
rng = np.random.RandomState(0)
for n in (6000, 8400, 12000, 16000):
    k = 256
    M = ctx.array(rng.normal(size=(k, n)))
    A0 = ctx.empty(n, n)
    ctx.syrk(n, k, 1.0, M, 0.0, A0)                               # low-rank + n I: SPD, well conditioned
    d = np.zeros(n); 
    h = A0.download(1, n)                                         # touch
    # add n to the diagonal on the host-free way: axpy on the strided diagonal is not available -> download/upload the diagonal only once
    diag = np.array([0.0])
    A_host = None
    ref = None
    if n <= 100:
        A_host = np.tril(A0.download()); A_host = A_host + np.tril(A_host, -1).T + n * np.eye(n)
        ref = np.linalg.cholesky(A_host)
        A0.upload(A_host)
    else:
        # large n: shift the diagonal by uploading a strided column would need the whole matrix; use identity nugget through assemble-free trick
        A_host = A0.download(); A_host[np.arange(n), np.arange(n)] += n; A0.upload(A_host); del A_host
    for label, keys in (('sequential        ', {20: 0}), ('pipelined c=32 LL ', {20: 100000, 13: 32, 18: 1}), ('pipelined c=48 LL ', {20: 100000, 13: 48, 18: 1}),
                        ('pipelined c=64 LL ', {20: 100000, 13: 64, 18: 1}), ('pipelined c=80 LL ', {20: 100000, 13: 80, 18: 1}), ('pipelined c=64 RL ', {20: 100000, 13: 64, 18: 0})):
        for kk, v in keys.items():
            ctx.lib.gpk_debug_set(kk, v)
        best = 1e9
        for rep in range(3):
            dA = A0.clone()
            ctx.synchronize(); ctx.timer_start(); info = ctx.potrf(dA); best = min(best, ctx.timer_stop())
            if rep < 2: dA.free()
        err = ''
        if ref is not None:
            err = 'err %.1e' % (np.max(np.abs(np.tril(dA.download()) - ref)) / np.max(np.abs(ref)))
        dA.free()
        print(f'n={n:6d} {label}: {best:8.3f} ms  {n ** 3 / 3 / best / 1e9:6.1f} TF/s info {info} {err}')
    A0.free(); M.free()
ctx.lib.gpk_debug_set(20, 0); ctx.lib.gpk_debug_set(13, 32); ctx.lib.gpk_debug_set(18, 1)

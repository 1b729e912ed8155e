#!/usr/bin/env python3
"""gpk_potrf at a few orders: right-looking on one stream (default) vs the two-partition forms (gpk_tune key 20 = largest order, key 54:
1 = right-looking with look-ahead (round 5), 0 = left-looking pipeline), with chain partitions of 32 and 64 CUs (key 13); result compared
with the default's factor.  python tools/potrf_modes_probe.py [orders ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
from src.sample_points import sampled_pts_rdm
ctx = gpk.Context(0, dev=True)                                  # (key 54 = 1, the look-ahead form, exists only in the development build)
orders = [int(a) for a in sys.argv[1:]] or [8400, 9600, 21000]
for N in orders:
    Nd = N * 10 // 21; Nb = N - 2 * Nd
    np.random.seed(0)
    Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]))
    N = 2 * Xd.shape[0] + Xb.shape[0]
    T = ctx.empty(N, N)
    ref = None
    for label, sets in (('one stream (default)', {20: 0}), ('look-ahead, 32-CU chain', {20: 10 ** 6, 54: 1, 13: 32}), ('look-ahead, 64-CU chain', {20: 10 ** 6, 54: 1, 13: 64}),
                        ('left-looking pipeline, 32', {20: 10 ** 6, 54: 0, 13: 32}), ('left-looking pipeline, 64', {20: 10 ** 6, 54: 0, 13: 64}), ('one stream again', {20: 0, 13: 32})):
        for k, v in sets.items():
            ctx.tune(k, v)
        best = 1e9
        for rep in range(3):
            ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-10, 'adaptive', out=T)
            ctx.timer_start(); info = ctx.potrf(T); ms = ctx.timer_stop(); best = min(best, ms)
        d = T.download(rows=min(N, 2048), row0=N - min(N, 2048))                # the last rows of the factor: they carry every update
        d = np.tril(d[:, :N], N - d.shape[0])
        if ref is None:
            ref = d
        dev = np.max(np.abs(d - ref)) / np.max(np.abs(ref))
        print(f'N={N} {label:28s}: {best:7.2f} ms, info {info}, {N**3/3/best/1e9:5.1f} TF/s, last rows vs default {dev:.1e}', flush=True)
    T.free()
ctx.tune(20, 0); ctx.tune(13, 32)

#!/usr/bin/env python3
"""Small Cholesky calls (development aid; run under `timeout`): n = 64, 100, 200, 513 against numpy."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
ctx = gpk.Context(0)
rng = np.random.RandomState(0)
for n in [int(a) for a in sys.argv[1:]] or [64, 100, 200, 513]:
    M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n)
    dA = ctx.array(A)
    t0 = time.time(); info = ctx.potrf(dA); dt = time.time() - t0
    L = np.tril(dA.download())
    print(f'n={n} info={info} {dt * 1e3:.1f} ms err={np.max(np.abs(L - np.linalg.cholesky(A))):.2e}', flush=True)

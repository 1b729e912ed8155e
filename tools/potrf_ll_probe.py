#!/usr/bin/env python3
"""Round-6 experiment (gpk_tune key 56): right-looking (default) against left-looking fused rank-64 work inside the 512-blocks of the
one-stream Cholesky, at the orders of the BASELINE workloads; min of 4 warm calls, factor compared bit for bit / to rounding."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import torch
import gpk
ctx = gpk.Context(0, dev=True)
for n in (4001, 6001, 8400, 9601, 10001, 21000):
    g = torch.Generator(device='cuda').manual_seed(n)
    M = torch.randn((n, 256), dtype=torch.float64, device='cuda', generator=g)
    A = (M @ M.T + n * torch.eye(n, dtype=torch.float64, device='cuda')).cpu().numpy()
    del M
    dA0 = ctx.array(A)
    row, ref = {}, None
    for name, k48, k56 in (('right_fused', 1, 0), ('left_fused', 2, 1), ('left_separate', 1, 1)):
        ctx.tune(48, k48); ctx.tune(56, k56)
        best = 1e9
        for rep in range(5):
            dA = dA0.clone()
            ctx.synchronize(); ctx.timer_start(); info = ctx.potrf(dA); ms = ctx.timer_stop()
            if rep:
                best = min(best, ms)
            if rep == 4 and n <= 10001:
                L = np.tril(dA.download())
                if ref is None:
                    ref = L
                row[name + '_reldiff'] = float(np.max(np.abs(L - ref)) / np.max(np.abs(ref)))
            dA.free()
        row[name] = round(best, 3)
    ctx.tune(48, 1); ctx.tune(56, 0)
    print(n, json.dumps(row), flush=True)
    dA0.free()
ctx.close()

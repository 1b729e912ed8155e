#!/usr/bin/env python3
"""Round 6: gpk_potrf at the orders of the BASELINE workloads (H of config 3 / 4, Theta of config 2, north-star H) against the outer
block width (gpk_tune key 51) and the fused / separate rank-64 schedule (key 48): min of 4 warm calls each, SPD test matrices."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk
import torch
ctx = gpk.Context(0, dev=True)
out = {}
for n in (4001, 6001, 8400, 9601, 10001):
    g = torch.Generator(device='cuda').manual_seed(n)
    M = torch.randn((n, 256), dtype=torch.float64, device='cuda', generator=g)
    A = (M @ M.T + n * torch.eye(n, dtype=torch.float64, device='cuda')).cpu().numpy()
    del M
    dA0 = ctx.array(A)
    row = {}
    for fused in (1, 0):
        for ob in (256, 384, 512, 768, 1024):
            ctx.tune(48, fused); ctx.tune(51, ob)
            best = 1e9
            for rep in range(5):
                dA = dA0.clone()
                ctx.synchronize(); ctx.timer_start(); info = ctx.potrf(dA); ms = ctx.timer_stop()
                if rep:
                    best = min(best, ms)
                dA.free()
            row[f'fused{fused}_ob{ob}'] = round(best, 3)
    ctx.tune(48, 1); ctx.tune(51, 512)
    out[n] = row
    print(n, json.dumps(row), flush=True)
    dA0.free()
ctx.close()

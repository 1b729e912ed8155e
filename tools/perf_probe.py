#!/usr/bin/env python3
"""Per-phase timings of the hot path on one GPU (development probe; bench.py is the contract benchmark)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk  # noqa: E402


def timed(ctx, fn, reps=3):
    fn(); ctx.synchronize()
    best = 1e30
    for _ in range(reps):
        ctx.timer_start(); fn(); best = min(best, ctx.timer_stop())
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--Nd', type=int, default=4000)
    ap.add_argument('--Nb', type=int, default=400)
    ap.add_argument('--nugget', type=float, default=1e-13)
    ap.add_argument('--steps', type=int, default=3)
    a = ap.parse_args()
    ctx = gpk.Context(0, dev=True)
    print(json.dumps(ctx.device_info()))
    print('ubench mfma f64 TF/s', ctx.ubench_mfma_f64(20000), ' hbm write GB/s', ctx.ubench_hbm_write(1 << 30, 10))
    Nd, Nb = a.Nd, a.Nb
    N, nz = 2 * Nd + Nb, Nd
    np.random.seed(0)
    Xd = np.random.uniform(0, 1, (Nd, 2)); Xb = np.random.uniform(0, 1, (Nb, 2))
    T = ctx.empty(N, N)
    ms = timed(ctx, lambda: ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, a.nugget, 'adaptive', out=T), 3)
    # (includes H2D of points + sync; kernel-only numbers come from rocprof)
    print(f'assemble N={N}: {ms:.3f} ms  -> {8.0 * N * N / ms / 1e6:.0f} GB/s (algorithmic 8N^2)')
    ctx.timer_start(); info = ctx.potrf(T); ms = ctx.timer_stop()
    print(f'potrf N={N}: {ms:.3f} ms info={info} -> {N ** 3 / 3 / ms / 1e9:.2f} TF/s')
    # standalone GEMM / SYRK / TRSM rates
    for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 2048), (4001, 4001, 8400)]:
        A = ctx.empty(k, m); B = ctx.empty(k, n); Cm = ctx.empty(m, n)
        A.upload(np.random.normal(size=(k, m))); B.upload(np.random.normal(size=(k, n)))
        ms = timed(ctx, lambda: ctx.gemm(1, 0, m, n, k, 1.0, A, B, 0.0, Cm))
        print(f'gemm TN {m}x{n}x{k}: {ms:.3f} ms -> {2.0 * m * n * k / ms / 1e9:.2f} TF/s')
        ms = timed(ctx, lambda: ctx.gemm(0, 1, m, n, k, 1.0, ctx_view(A, m, k), ctx_view(B, n, k), 0.0, Cm)) if False else ms
        A.free(); B.free(); Cm.free()
    S = ctx.empty(N, nz + 1); S.upload(np.random.normal(size=(N, nz + 1)))
    ms = timed(ctx, lambda: ctx.trsm(T, S), 2)
    print(f'trsm N={N} nrhs={nz + 1}: {ms:.3f} ms -> {float(N) * N * (nz + 1) / ms / 1e9:.2f} TF/s')
    H = ctx.empty(nz + 1, nz + 1)
    ms = timed(ctx, lambda: ctx.syrk(nz + 1, N, 1.0, S, 0.0, H))
    print(f'syrk n={nz + 1} k={N}: {ms:.3f} ms -> {float(N) * (nz + 1) ** 2 / ms / 1e9:.2f} TF/s')
    S.free(); H.free()
    # GN steps
    from_truth = lambda x1, x2: np.sin(np.pi * x1) * np.sin(np.pi * x2) + 2 * np.sin(4 * np.pi * x1) * np.sin(4 * np.pi * x2)
    f = 2 * np.pi ** 2 * np.sin(np.pi * Xd[:, 0]) * np.sin(np.pi * Xd[:, 1]) + 64 * np.pi ** 2 * np.sin(4 * np.pi * Xd[:, 0]) * np.sin(4 * np.pi * Xd[:, 1]) + from_truth(Xd[:, 0], Xd[:, 1]) ** 3
    g = from_truth(Xb[:, 0], Xb[:, 1])
    prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
    z = ctx.array(np.random.normal(size=Nd))
    for it in range(a.steps):
        t0 = time.time(); loss, info = ctx.gn_step(prob, z); dt = time.time() - t0
        flops = float(N) * N * nz + float(N) * nz * nz + nz ** 3 / 3.0
        print(f'gn_step {it}: {dt * 1e3:.2f} ms loss_in={loss:.6e} info={info} -> {flops / dt / 1e12:.2f} TF/s (F1 flops)')
    t0 = time.time(); lf = ctx.gn_loss(prob, z); print(f'gn_loss: {(time.time() - t0) * 1e3:.2f} ms loss={lf:.6e}')
    sol = z.download()
    print('collocation L2 err', np.sqrt(np.mean((from_truth(Xd[:, 0], Xd[:, 1]) - sol) ** 2)))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Hyper-parameter sweep (kernel length scale sigma x nugget) over ONE collocation set for the nonlinear elliptic problem.

The reference's notebooks hand-tune sigma (e.g. sigma = 0.15878296 in Nonlinear_Elliptic_Equation.ipynb cell 15); every
(sigma, nugget) pair is an independent assemble -> Cholesky -> Gauss-Newton -> test solve, so the grid is sharded over
ranks with NO data-path collective (replicas): rank r takes the pairs r, r+P, r+2P, ... and rank 0 gathers
the small result records.

    python sweep.py --sigmas 0.1 0.15 0.2 0.25 --nuggets 1e-8 1e-10 1e-12
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 sweep.py --sigmas ... --nuggets ...
"""
import argparse
import itertools
import json
import os
import sys

import numpy as onp

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def shard(items, rank, world):
    """round-robin share of `items` for `rank` (deterministic, covers every item exactly once over all ranks)"""
    return [it for k, it in enumerate(items) if k % world == rank]


def solve_one(X_domain, X_boundary, sigma, nugget, GNsteps, alpha, m, init_sol, X_test):
    from src.PDEs import Nonlinear_elliptic2d
    pi = onp.pi
    u = lambda x1, x2: onp.sin(pi * x1) * onp.sin(pi * x2) + 2 * onp.sin(4 * pi * x1) * onp.sin(4 * pi * x2)
    f = lambda x1, x2: (2 * pi ** 2 * onp.sin(pi * x1) * onp.sin(pi * x2)
                        + 64 * pi ** 2 * onp.sin(4 * pi * x1) * onp.sin(4 * pi * x2) + alpha * u(x1, x2) ** m)
    eq = Nonlinear_elliptic2d(alpha=alpha, m=m, bdy=u, rhs=f)
    eq.get_sampled_points(X_domain, X_boundary)
    eq.Gram_matrix(kernel='Gaussian', kernel_parameter=sigma, nugget=nugget, nugget_type='adaptive')
    eq.Gram_Cholesky()
    rec = {'sigma': sigma, 'nugget': nugget, 'chol_info': int(eq.chol_info)}
    if eq.chol_info == 0:
        onp.random.seed(12345)                                   # same initial guess for every pair
        eq.GN_method(max_iter=GNsteps, step_size=1, initial_sol='rdm', print_hist=False)
        eq.extend_sol(X_test)
        rec['loss'] = float(eq.loss_hist[-1])
        rec['pts_L2_err'] = float(onp.sqrt(onp.mean((u(X_domain[:, 0], X_domain[:, 1]) - eq.sol_sampled_pts) ** 2)))
        rec['test_L2_err'] = float(onp.sqrt(onp.mean((u(X_test[:, 0], X_test[:, 1]) - eq.extended_sol) ** 2)))
    eq._drop_device_state()
    return rec


def main(argv=None):
    ap = argparse.ArgumentParser(description='sigma x nugget sweep, replicas sharded over ranks')
    ap.add_argument('--sigmas', type=float, nargs='+', default=[0.1, 0.15, 0.2, 0.25])
    ap.add_argument('--nuggets', type=float, nargs='+', default=[1e-8, 1e-10, 1e-12])
    ap.add_argument('--N_domain', type=int, default=900)
    ap.add_argument('--N_boundary', type=int, default=124)
    ap.add_argument('--GNsteps', type=int, default=5)
    ap.add_argument('--alpha', type=float, default=1.0)
    ap.add_argument('--m', type=float, default=3.0)
    ap.add_argument('--randomseed', type=int, default=0)
    a = ap.parse_args(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo')                          # only small Python records are exchanged
    from src.sample_points import sampled_pts_rdm
    onp.random.seed(a.randomseed)                                # identical points on every rank
    Xd, Xb = sampled_pts_rdm(a.N_domain, a.N_boundary, onp.array([[0, 1], [0, 1]]))
    xx = onp.linspace(0, 1, 60)
    XX, YY = onp.meshgrid(xx, xx)
    Xt = onp.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)
    grid = list(itertools.product(a.sigmas, a.nuggets))
    mine = [solve_one(Xd, Xb, s, n, a.GNsteps, a.alpha, a.m, None, Xt) for (s, n) in shard(grid, rank, world)]
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        dist.destroy_process_group()
        if rank != 0:
            return None
        mine = [r for part in gathered for r in part]
    mine.sort(key=lambda r: (r['sigma'], r['nugget']))
    for r in mine:
        print(json.dumps(r))
    best = min((r for r in mine if 'test_L2_err' in r), key=lambda r: r['test_L2_err'], default=None)
    if best:
        print(f"[Sweep] best test L2 error {best['test_L2_err']:.3e} at sigma = {best['sigma']}, nugget = {best['nugget']}")
    return mine


if __name__ == '__main__':
    main()

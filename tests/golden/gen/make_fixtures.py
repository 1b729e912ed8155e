#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the REFERENCE's own code (authoring container only).

    python -B tests/golden/gen/make_fixtures.py            # needs /root/reference

* src/sample_points.py imports only numpy: loaded directly by path, no stand-in involved.
* every other src/ module starts with `import jax`; JAX is not installed here and cannot be (no network), so
  those modules are executed UNCHANGED on the torch.func stand-in in tests/golden/gen/jax_shim/ (see its
  docstring: jit = identity, fp64 torch CPU).  The vectors are therefore "reference source, non-JAX runtime";
  the notebook known-answer tests (tests/test_oracle_kat.py) are the pin that involves real JAX output.
* only DATA is stored (inputs and outputs as arrays); no reference source text goes into the fixtures.
"""
import argparse
import importlib.util
import os
import sys

sys.dont_write_bytecode = True                      # never write __pycache__ into the read-only reference tree
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.dirname(HERE)
REF = '/root/reference'

import numpy as np  # noqa: E402


def load_reference_sampler():
    spec = importlib.util.spec_from_file_location('ref_sample_points', os.path.join(REF, 'src', 'sample_points.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_sampling():
    sp = load_reference_sampler()
    out = {}
    cases = [('static_5_8', 5, 8, [[0, 1], [0, 1]], False, 0),
             ('static_900_124', 900, 124, [[0, 1], [0, 1]], False, 0),
             ('static_37_14', 37, 14, [[-1, 2], [0.5, 3]], False, 7),
             ('time_1000_200', 1000, 200, [[0, 1], [-1, 1]], True, 0),
             ('time_2000_400', 2000, 400, [[0, 1], [-1, 1]], True, 0),
             ('time_31_10', 31, 10, [[0, 1], [-1, 1]], True, 3)]
    for name, nd, nb, dom, td, seed in cases:
        np.random.seed(seed)
        Xd, Xb = sp.sampled_pts_rdm(nd, nb, np.array(dom), time_dependent=td)
        tail = np.random.uniform(0, 1, 3)              # pins how much of the stream was consumed
        out[f'{name}__Xd'] = Xd; out[f'{name}__Xb'] = Xb; out[f'{name}__tail'] = tail
        out[f'{name}__args'] = np.array([nd, nb, dom[0][0], dom[0][1], dom[1][0], dom[1][1], float(td), seed])
    for name, nd, nb, dom, td in [('grid_static_900_124', 900, 124, [[0, 1], [0, 1]], False),
                                  ('grid_static_50_20', 50, 20, [[0, 2], [-1, 1]], False),
                                  ('grid_time_1000_200', 1000, 200, [[0, 1], [-1, 1]], True),
                                  ('grid_time_40_9', 40, 9, [[0, 1], [-1, 1]], True)]:
        Xd, Xb = sp.sampled_pts_grid(nd, nb, np.array(dom), time_dependent=td)
        out[f'{name}__Xd'] = Xd; out[f'{name}__Xb'] = Xb
        out[f'{name}__args'] = np.array([nd, nb, dom[0][0], dom[0][1], dom[1][0], dom[1][1], float(td), -1])
    np.savez_compressed(os.path.join(GOLDEN, 'sampling.npz'), **out)
    print('sampling.npz', len(out))


def import_reference_src():
    sys.path.insert(0, os.path.join(HERE, 'jax_shim'))
    sys.path.insert(1, REF)
    import matplotlib
    matplotlib.use('Agg')
    from src import Gram_matrice, PDEs, InverseProblems, solver  # noqa: F401
    return Gram_matrice, PDEs, InverseProblems, solver


THETA_CASES = [
    ('elliptic_gauss', 'Nonlinear_elliptic', 'Gaussian', 0.2, False),
    ('elliptic_aniso', 'Nonlinear_elliptic', 'anisotropic_Gaussian', [0.3, 0.4], False),
    ('burgers_aniso', 'Burgers', 'anisotropic_Gaussian', [0.3, 0.05], True),
    ('burgers_gauss', 'Burgers', 'Gaussian', 0.25, True),
    ('eikonal_gauss', 'Eikonal', 'Gaussian', 0.2, False),
    ('eikonal_aniso', 'Eikonal', 'anisotropic_Gaussian', [0.5, 0.3], False),
    ('darcy_gauss', 'Darcy_flow2d', 'Gaussian', 0.2, False),
]


def make_theta(Gram):
    sp = load_reference_sampler()
    out = {}
    for name, eqn, kernel, kp, td in THETA_CASES:
        np.random.seed(11)
        dom = np.array([[0, 1], [-1, 1]]) if td else np.array([[0, 1], [0, 1]])
        Xd, Xb = sp.sampled_pts_rdm(37, 13, dom, time_dependent=td)
        Xt = np.random.uniform(dom[:, 0], dom[:, 1], (9, 2))
        T = Gram.Gram_matrix_assembly(Xd, Xb, eqn=eqn, kernel=kernel, kernel_parameter=kp)
        Tt = Gram.construct_Theta_test(Xt, Xd, Xb, eqn=eqn, kernel=kernel, kernel_parameter=kp)
        out[f'{name}__Xd'] = Xd; out[f'{name}__Xb'] = Xb; out[f'{name}__Xt'] = Xt
        out[f'{name}__kp'] = np.atleast_1d(np.asarray(kp, dtype=float))
        if eqn == 'Darcy_flow2d':
            out[f'{name}__Theta_u'] = np.asarray(T[0]); out[f'{name}__Theta_a'] = np.asarray(T[1])
            out[f'{name}__Theta_u_test'] = np.asarray(Tt[0]); out[f'{name}__Theta_a_test'] = np.asarray(Tt[1])
        else:
            out[f'{name}__Theta'] = np.asarray(T); out[f'{name}__Theta_test'] = np.asarray(Tt)
    np.savez_compressed(os.path.join(GOLDEN, 'theta_small.npz'), **out)
    print('theta_small.npz', len(out))


def _cfg(**kw):
    return argparse.Namespace(**kw)


def _elliptic_callbacks(alpha, m):
    import jax.numpy as jnp
    from jax import grad

    def u(x1, x2):
        return jnp.sin(jnp.pi * x1) * jnp.sin(jnp.pi * x2) + 2 * jnp.sin(4 * jnp.pi * x1) * jnp.sin(4 * jnp.pi * x2)

    def f(x1, x2):      # same construction as main_NonLinElliptic2d.py:63-64
        return -grad(grad(u, 0), 0)(x1, x2) - grad(grad(u, 1), 1)(x1, x2) + alpha * (u(x1, x2) ** m)
    return u, f


def _grid(n, dom):
    xx = np.linspace(dom[0][0], dom[0][1], n); yy = np.linspace(dom[1][0], dom[1][1], n)
    XX, YY = np.meshgrid(xx, yy)
    return np.concatenate((XX.reshape(-1, 1), YY.reshape(-1, 1)), axis=1)


def _common(eq, out, pre):
    for k in ('X_domain', 'X_boundary', 'rhs_f', 'bdy_g', 'init_sol'):
        out[f'{pre}__{k}'] = np.asarray(getattr(eq, k), dtype=float)
    out[f'{pre}__loss_hist'] = np.array([float(v) for v in eq.loss_hist])


def make_solves(solver_mod):
    import jax.numpy as jnp
    solver_GP = solver_mod.solver_GP
    out = {}

    # ---- Nonlinear elliptic, two sizes: a well-conditioned one and BASELINE config 1 ----
    for pre, nd, nb, nug, steps, seed, ntest in [('elliptic_small', 300, 60, 1e-8, 4, 5, 15),
                                                 ('elliptic_c1', 900, 124, 1e-13, 4, 0, 60)]:
        cfg = _cfg(alpha=1.0, m=3.0, kernel='Gaussian', kernel_parameter=0.2, nugget=nug, nugget_type='adaptive',
                   GNsteps=steps, step_size=1, initial_sol='rdm', print_hist=False)
        np.random.seed(seed)
        s = solver_GP(cfg, PDE_type='Nonlinear_elliptic')
        u, f = _elliptic_callbacks(cfg.alpha, cfg.m)
        dom = np.array([[0, 1], [0, 1]])
        s.set_equation(bdy=u, rhs=f, domain=dom, print_option=False)
        s.auto_sample(nd, nb, sampled_type='random', print_option=False)
        s.solve(method='elimination', print_option=False)
        Xt = _grid(ntest, dom)
        s.test(Xt, print_option=False)
        _common(s.eqn, out, pre)
        out[f'{pre}__params'] = np.array([cfg.alpha, cfg.m, 0.2, nug, steps, seed])
        out[f'{pre}__ratio'] = np.atleast_1d(np.asarray(s.eqn.ratio, dtype=float))
        out[f'{pre}__sol'] = np.asarray(s.eqn.sol_sampled_pts, dtype=float)
        out[f'{pre}__sol_vec'] = np.asarray(s.eqn.sol_vec, dtype=float)
        out[f'{pre}__X_test'] = Xt
        out[f'{pre}__extended_sol'] = np.asarray(s.eqn.extended_sol, dtype=float)
        print(pre, 'loss', out[f'{pre}__loss_hist'])

    # ---- relaxed elliptic ----
    pre = 'elliptic_relaxed'
    cfg = _cfg(alpha=1.0, m=3.0, kernel='Gaussian', kernel_parameter=0.2, nugget=1e-8, nugget_type='adaptive',
               GNsteps=3, step_size=1, initial_sol='rdm', print_hist=False)
    np.random.seed(6)
    s = solver_GP(cfg, PDE_type='Nonlinear_elliptic')
    u, f = _elliptic_callbacks(cfg.alpha, cfg.m)
    dom = np.array([[0, 1], [0, 1]])
    s.set_equation(bdy=u, rhs=f, domain=dom, print_option=False)
    s.auto_sample(120, 40, sampled_type='random', print_option=False)
    s.solve(method='relaxation', pen_lambda=1e-6, print_option=False)
    _common(s.eqn, out, pre)
    out[f'{pre}__params'] = np.array([1.0, 3.0, 0.2, 1e-8, 3, 6, 1e-6])
    out[f'{pre}__sol'] = np.asarray(s.eqn.sol_sampled_pts, dtype=float)
    out[f'{pre}__sol_vec'] = np.asarray(s.eqn.sol_vec, dtype=float)
    print(pre, 'loss', out[f'{pre}__loss_hist'])

    # ---- Burgers (anisotropic kernel), as main_Burgers1d.py:65-80 ----
    pre = 'burgers_small'
    cfg = _cfg(alpha=1.0, nu=0.02, kernel='anisotropic_Gaussian', kernel_parameter=[0.3, 0.05], nugget=1e-5,
               nugget_type='adaptive', GNsteps=6, step_size=1, initial_sol='rdm', print_hist=False)
    np.random.seed(0)
    s = solver_GP(cfg, PDE_type='Burgers')

    def ub(x1, x2):
        return -jnp.sin(jnp.pi * x2) * (x1 == 0) + 0 * (x2 == 0)

    def fb(x1, x2):
        return 0
    dom = np.array([[0, 1], [-1, 1]])
    s.set_equation(bdy=ub, rhs=fb, domain=dom, print_option=False)
    s.auto_sample(200, 60, sampled_type='random', print_option=False)
    s.solve(print_option=False)
    Xt = _grid(15, dom)
    s.test(Xt, print_option=False)
    _common(s.eqn, out, pre)
    out[f'{pre}__params'] = np.array([cfg.alpha, cfg.nu, 0.3, 0.05, cfg.nugget, cfg.GNsteps, 0])
    out[f'{pre}__ratio'] = np.asarray([float(r) for r in s.eqn.ratio])
    out[f'{pre}__sol'] = np.asarray(s.eqn.sol_sampled_pts, dtype=float)
    out[f'{pre}__sol_vec'] = np.asarray(s.eqn.sol_vec, dtype=float)
    out[f'{pre}__X_test'] = Xt
    out[f'{pre}__extended_sol'] = np.asarray(s.eqn.extended_sol, dtype=float)
    print(pre, 'loss', out[f'{pre}__loss_hist'])

    # ---- Eikonal, as main_Eikonal2d.py:53-66 ----
    pre = 'eikonal_small'
    cfg = _cfg(eps=1e-1, kernel='Gaussian', kernel_parameter=0.2, nugget=1e-5, nugget_type='adaptive',
               GNsteps=6, step_size=1, initial_sol='zero', print_hist=False)
    np.random.seed(2)
    s = solver_GP(cfg, PDE_type='Eikonal')
    dom = np.array([[0, 1], [0, 1]])
    s.set_equation(bdy=lambda x1, x2: 0, rhs=lambda x1, x2: 1, domain=dom, print_option=False)
    s.auto_sample(200, 48, sampled_type='random', print_option=False)
    s.solve(print_option=False)
    Xt = _grid(15, dom)
    s.test(Xt, print_option=False)
    _common(s.eqn, out, pre)
    out[f'{pre}__params'] = np.array([cfg.eps, 0.2, cfg.nugget, cfg.GNsteps, 2])
    out[f'{pre}__sol'] = np.asarray(s.eqn.sol_sampled_pts, dtype=float)
    out[f'{pre}__sol_vec'] = np.asarray(s.eqn.sol_vec, dtype=float)
    out[f'{pre}__X_test'] = Xt
    out[f'{pre}__extended_sol'] = np.asarray(s.eqn.extended_sol, dtype=float)
    print(pre, 'loss', out[f'{pre}__loss_hist'])

    # ---- Darcy inverse problem (synthetic observations: smooth function + noise through get_observed_data) ----
    pre = 'darcy_small'
    cfg = _cfg(kernel='Gaussian', kernel_parameter=0.2, nugget=1e-6, nugget_type='adaptive',
               GNsteps=6, step_size=1, initial_sol='rdm', print_hist=False)
    np.random.seed(9999)
    s = solver_GP(cfg, PDE_type='Darcy_flow2d')
    s.set_equation(bdy=lambda x1, x2: 0, rhs=lambda x1, x2: 1, domain=dom, print_option=False)
    s.auto_sample_IP(150, 40, 20, sampled_type='random', print_option=False)
    Xdat = s.eqn.X_data
    data_clean = 0.02 * np.sin(np.pi * Xdat[:, 0]) * np.sin(np.pi * Xdat[:, 1])
    s.get_observed_data(data_clean, 1e-3, print_option=False)
    s.eqn.data_u = np.asarray(s.eqn.data_u).view(type(jnp.zeros(1)))
    s.solve(print_option=False)
    Xt = _grid(12, dom)
    s.test(Xt, print_option=False)
    _common(s.eqn, out, pre)
    out[f'{pre}__params'] = np.array([0.2, cfg.nugget, cfg.GNsteps, 9999, 20, 1e-3])
    out[f'{pre}__data_clean'] = data_clean
    out[f'{pre}__data_u'] = np.asarray(s.eqn.data_u, dtype=float)
    out[f'{pre}__sol_vec_a'] = np.asarray(s.eqn.sol_vec_a, dtype=float)
    out[f'{pre}__sol_vec_u'] = np.asarray(s.eqn.sol_vec_u, dtype=float)
    out[f'{pre}__X_test'] = Xt
    out[f'{pre}__extended_sol_a'] = np.asarray(s.eqn.extended_sol_a, dtype=float)
    out[f'{pre}__extended_sol_u'] = np.asarray(s.eqn.extended_sol_u, dtype=float)
    print(pre, 'loss', out[f'{pre}__loss_hist'])

    np.savez_compressed(os.path.join(GOLDEN, 'solves.npz'), **out)
    print('solves.npz', len(out))


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit('needs /root/reference (authoring container)')
    which = set(sys.argv[1:]) or {'sampling', 'theta', 'solves'}
    if 'sampling' in which:
        make_sampling()
    if which & {'theta', 'solves'}:
        Gram, PDEs, IP, solver_mod = import_reference_src()
        if 'theta' in which:
            make_theta(Gram)
        if 'solves' in which:
            make_solves(solver_mod)

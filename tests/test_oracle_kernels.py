"""Closed-form derivative kernels of the oracle vs symbolic nested differentiation.

Each method of src/kernels.py is a nest of `jax.grad` calls on the scalar kappa; the same nests are written
here with sympy.diff (argument indices 0..3 = x1,x2,y1,y2 as in `grad(f, argnum)`), lambdified and compared
with oracle.gp_oracle.deriv_kernel on random pairs.  Independent of JAX and of any stand-in for it.
"""
import numpy as np
import pytest
import sympy as sp

from oracle import gp_oracle as O

x1, x2, y1, y2 = sp.symbols('x1 x2 y1 y2', real=True)
ARGS = (x1, x2, y1, y2)


def _g(expr, *argnums):
    """grad(grad(expr, a0), a1)... in application order."""
    for a in argnums:
        expr = sp.diff(expr, ARGS[a])
    return expr


def _nests(kappa):
    """The nests of src/kernels.py:15-89 (Gaussian) / :102-178 (anisotropic), method by method."""
    dyk = _g(kappa, 2, 2) + _g(kappa, 3, 3)            # Delta_y_kappa   :72-75
    dxk = _g(kappa, 0, 0) + _g(kappa, 1, 1)            # Delta_x_kappa   :67-70
    return {
        'kappa': kappa,
        'D_x1_kappa': _g(kappa, 0), 'D_x2_kappa': _g(kappa, 1), 'DD_x2_kappa': _g(kappa, 1, 1),
        'D_y1_kappa': _g(kappa, 2), 'D_y2_kappa': _g(kappa, 3), 'DD_y2_kappa': _g(kappa, 3, 3),
        'D_x1_D_y1_kappa': _g(kappa, 0, 2), 'D_x1_D_y2_kappa': _g(kappa, 0, 3),
        'D_x1_DD_y2_kappa': _g(kappa, 0, 3, 3), 'D_x2_D_y2_kappa': _g(kappa, 1, 3),
        'D_x2_D_y1_kappa': _g(_g(kappa, 1), 2), 'D_x2_DD_y2_kappa': _g(kappa, 1, 3, 3),
        'DD_x2_DD_y2_kappa': _g(kappa, 1, 1, 3, 3),
        'Delta_x_kappa': dxk, 'Delta_y_kappa': dyk,
        'Delta_x_Delta_y_kappa': _g(dyk, 0, 0) + _g(dyk, 1, 1),       # :77-80
        'Delta_x_y_kappa': _g(dyk, 0, 0) + _g(dyk, 1, 1),             # aniso :163-166
        'Delta_x_D_y1_kappa': _g(dxk, 2), 'Delta_x_D_y2_kappa': _g(dxk, 3),
    }


CASES = [
    ('Gaussian', 0.2, lambda s: sp.exp(-(1 / (2 * s ** 2)) * ((x1 - y1) ** 2 + (x2 - y2) ** 2))),
    ('Gaussian', 0.37, lambda s: sp.exp(-(1 / (2 * s ** 2)) * ((x1 - y1) ** 2 + (x2 - y2) ** 2))),
    ('anisotropic_Gaussian', (0.3, 0.05), lambda s: sp.exp(-(((x1 - y1) / s[0]) ** 2 + ((x2 - y2) / s[1]) ** 2))),
    ('anisotropic_Gaussian', (1 / 3, 1 / 20), lambda s: sp.exp(-(((x1 - y1) / s[0]) ** 2 + ((x2 - y2) / s[1]) ** 2))),
]


@pytest.mark.parametrize('kernel,param,expr', CASES)
def test_closed_forms_match_nested_differentiation(kernel, param, expr):
    sym_param = sp.Rational(str(param)) if np.isscalar(param) else tuple(sp.nsimplify(p) for p in param)
    nests = _nests(expr(sym_param))
    rng = np.random.RandomState(1)
    lo, hi = (0.0, 1.0)
    X = rng.uniform(lo, hi, (400, 2)); Y = rng.uniform(lo, hi, (400, 2))
    if kernel == 'anisotropic_Gaussian':                 # keep exp() away from total underflow
        Y = X + rng.uniform(-1, 1, (400, 2)) * np.array([3 * param[0], 3 * param[1]])
    for name, e in nests.items():
        f = sp.lambdify(ARGS, e, modules='numpy')
        want = np.asarray(f(X[:, 0], X[:, 1], Y[:, 0], Y[:, 1]), dtype=np.float64) * np.ones(400)
        got = O.deriv_kernel(name, X[:, 0], X[:, 1], Y[:, 0], Y[:, 1], kernel, param)
        scale = np.max(np.abs(want)) + 1e-300
        assert np.max(np.abs(got - want)) <= 2e-13 * scale, (kernel, param, name)


def test_method_inventory():
    """19 distinct names on the Gaussian class + the anisotropic duplicate Delta_x_y_kappa."""
    assert len(O.KERNEL_METHODS) == 20
    assert 'Delta_x_y_kappa' in O.KERNEL_METHODS


def test_diagonal_values():
    """Values at d=0 (SURVEY §8a-K last column) -- these make the adaptive-nugget trace ratios analytic."""
    s = 0.2
    p = 1 / s ** 2
    z = np.zeros(1)
    v = lambda n: O.deriv_kernel(n, z, z, z, z, 'Gaussian', s)[0]
    assert v('kappa') == 1.0
    assert v('D_x1_D_y1_kappa') == pytest.approx(p, rel=1e-15)
    assert v('DD_x2_DD_y2_kappa') == pytest.approx(3 * p * p, rel=1e-15)
    assert v('Delta_x_Delta_y_kappa') == pytest.approx(8 / s ** 4, rel=1e-15)
    assert v('Delta_x_kappa') == pytest.approx(-2 * p, rel=1e-15)
    for n in ('D_x1_kappa', 'D_y2_kappa', 'D_x1_D_y2_kappa', 'D_x2_DD_y2_kappa', 'Delta_x_D_y1_kappa'):
        assert v(n) == 0.0

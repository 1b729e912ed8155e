"""Gaussian and anisotropic-Gaussian kernels with the method names of the reference's src/kernels.py.

The reference obtains every derivative by nested jax.grad of the scalar kappa (src/kernels.py:15-89, 102-178).  Here each
method is the closed form  d_x^alpha d_y^beta kappa = (-1)^{|alpha|} h_{a1+b1}(p1, x1-y1) h_{a2+b2}(p2, x2-y2) kappa
with the Hermite factors h0..h4 of exp(-p d^2/2); the same formula is what the HIP Gram evaluator
(csrc/gpk_assemble.hip) computes per point pair.  Methods accept scalars or numpy arrays.
"""
import numpy as np


def _hermite(p, d):
    q = p * d
    q2 = q * q
    return (1.0, q, q2 - p, q * (q2 - 3.0 * p), q2 * (q2 - 6.0 * p) + 3.0 * p * p)


# functionals as lists of derivative multi-indices (order in x1/y1, order in x2/y2)
_ID, _D1, _D2, _DD2, _LAP = [(0, 0)], [(1, 0)], [(0, 1)], [(0, 2)], [(2, 0), (0, 2)]

# method name -> (functional applied in x, functional applied in y)
_METHODS = {
    'kappa': (_ID, _ID),
    'D_x1_kappa': (_D1, _ID), 'D_x2_kappa': (_D2, _ID), 'DD_x2_kappa': (_DD2, _ID),
    'D_y1_kappa': (_ID, _D1), 'D_y2_kappa': (_ID, _D2), 'DD_y2_kappa': (_ID, _DD2),
    'D_x1_D_y1_kappa': (_D1, _D1), 'D_x1_D_y2_kappa': (_D1, _D2), 'D_x1_DD_y2_kappa': (_D1, _DD2),
    'D_x2_D_y2_kappa': (_D2, _D2), 'D_x2_D_y1_kappa': (_D2, _D1), 'D_x2_DD_y2_kappa': (_D2, _DD2),
    'DD_x2_DD_y2_kappa': (_DD2, _DD2),
    'Delta_x_kappa': (_LAP, _ID), 'Delta_y_kappa': (_ID, _LAP), 'Delta_x_Delta_y_kappa': (_LAP, _LAP),
    'Delta_x_D_y1_kappa': (_LAP, _D1), 'Delta_x_D_y2_kappa': (_LAP, _D2),
}


class _ClosedFormKernel(object):
    def __init__(self):
        pass

    def _precisions(self, sigma):
        raise NotImplementedError

    def _eval(self, fx, fy, x1, x2, y1, y2, sigma):
        p1, p2 = self._precisions(sigma)
        d1 = np.asarray(x1, dtype=np.float64) - np.asarray(y1, dtype=np.float64)
        d2 = np.asarray(x2, dtype=np.float64) - np.asarray(y2, dtype=np.float64)
        a, b = _hermite(p1, d1), _hermite(p2, d2)
        total = 0.0
        for (a1, a2) in fx:
            for (b1, b2) in fy:
                term = a[a1 + b1] * b[a2 + b2]
                total = total - term if (a1 + a2) & 1 else total + term
        return total * np.exp(-0.5 * (p1 * d1 * d1 + p2 * d2 * d2))


def _make(name):
    fx, fy = _METHODS[name]

    def method(self, x1, x2, y1, y2, sigma):
        return self._eval(fx, fy, x1, x2, y1, y2, sigma)
    method.__name__ = name
    return method


for _n in _METHODS:
    setattr(_ClosedFormKernel, _n, _make(_n))


class Gaussian_kernel(_ClosedFormKernel):
    """kappa = exp(-((x1-y1)^2 + (x2-y2)^2) / (2 sigma^2))   (reference src/kernels.py:8-89)"""

    def _precisions(self, sigma):
        p = 1.0 / (float(sigma) ** 2)
        return p, p


class Anisotropic_Gaussian_kernel(_ClosedFormKernel):
    """kappa = exp(-((x1-y1)/sigma[0])^2 - ((x2-y2)/sigma[1])^2)   (reference src/kernels.py:91-179; no factor 1/2)"""

    def _precisions(self, sigma):
        return 2.0 / (float(sigma[0]) ** 2), 2.0 / (float(sigma[1]) ** 2)

    Delta_x_y_kappa = _make('Delta_x_Delta_y_kappa')      # duplicate of the reference class (src/kernels.py:163-166)

"""Gram_matrix_assembly / construct_Theta_test with the reference's signatures (src/Gram_matrice.py:11,190), evaluated
by the fused HIP block evaluator (gpk_assemble / gpk_assemble_test) and returned as numpy arrays."""
import numpy as onp

from ._runtime import get_context

_LAYOUTS = {'Nonlinear_elliptic': ('Nonlinear_elliptic',), 'Burgers': ('Burgers',), 'Eikonal': ('Eikonal',),
            'Darcy_flow2d': ('Darcy_u', 'Darcy_a')}
_KERNELS = ('Gaussian', 'anisotropic_Gaussian')


def Gram_matrix_assembly(X_domain, X_boundary, eqn='Nonlinear_elliptic', kernel='Gaussian', kernel_parameter=0.2):
    if eqn not in _LAYOUTS:
        return None                      # the reference falls through its if/elif chain and returns None
    if kernel not in _KERNELS:
        raise UnboundLocalError("local variable 'K' referenced before assignment")   # reference behaviour (:36-39)
    ctx = get_context()
    out = []
    for layout in _LAYOUTS[eqn]:
        T, _ = ctx.assemble(layout, kernel, kernel_parameter, onp.asarray(X_domain), onp.asarray(X_boundary))
        out.append(T.download())
        T.free()
    return out[0] if len(out) == 1 else tuple(out)


def construct_Theta_test(X_test, X_domain, X_boundary, eqn='Nonlinear_elliptic', kernel='Gaussian', kernel_parameter=0.2):
    if eqn not in _LAYOUTS:
        return None
    if kernel not in _KERNELS:
        raise UnboundLocalError("local variable 'K' referenced before assignment")
    ctx = get_context()
    out = []
    for layout in _LAYOUTS[eqn]:
        T = ctx.assemble_test(layout, kernel, kernel_parameter, onp.asarray(X_test), onp.asarray(X_domain), onp.asarray(X_boundary))
        out.append(T.download())
        T.free()
    return out[0] if len(out) == 1 else tuple(out)

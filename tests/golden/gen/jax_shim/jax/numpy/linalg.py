"""jax.numpy.linalg stand-in.  cholesky returns NaNs on failure (JAX semantics); solve is a GENERAL solve."""
import numpy as _np
import torch as _torch
from . import _is_t, _t, _v


def cholesky(a):
    if _is_t(a):
        return _torch.linalg.cholesky(a)
    try:
        return _v(_np.linalg.cholesky(_np.asarray(a)))
    except _np.linalg.LinAlgError:
        return _v(_np.full_like(_np.asarray(a), _np.nan))


def solve(a, b):
    if _is_t(a) or _is_t(b):
        return _torch.linalg.solve(_t(a), _t(b))
    return _v(_np.linalg.solve(_np.asarray(a), _np.asarray(b)))

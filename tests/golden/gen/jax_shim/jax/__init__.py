"""Minimal stand-in for the `jax` package, built on torch.func (CPU, float64).

PURPOSE: fixture generation ONLY (tests/golden/gen/make_fixtures.py), in the authoring container where
/root/reference exists and JAX does not.  It lets the reference's own src/*.py execute unchanged so that their
outputs can be stored as golden vectors.  It is NOT JAX: jit is the identity, XLA is not involved, the numbers
are torch CPU fp64.  Nothing in the product or in the GPU-side tests imports it.
"""
import numpy as _np
import torch as _torch
from torch import func as _F

_torch.set_default_dtype(_torch.float64)


class _Config:
    def update(self, *a, **k):
        pass


config = _Config()


def _is_t(x):
    return isinstance(x, _torch.Tensor)


def _to_t(x):
    if _is_t(x):
        return x
    return _torch.as_tensor(_np.asarray(x, dtype=_np.float64))


def _to_np(x):
    from .numpy import Arr
    if _is_t(x):
        return _np.asarray(x.detach().cpu().numpy()).view(Arr)
    if isinstance(x, (tuple, list)):
        return type(x)(_to_np(v) for v in x)
    return x


def _coerce_out(fn):
    def g(*a, **k):
        out = fn(*a, **k)
        if not _is_t(out):
            out = _torch.as_tensor(float(out), dtype=_torch.float64)
        return out
    return g


def jit(fun=None, static_argnums=None, **kw):
    if fun is None:
        return lambda f: f
    return fun


def _wrap_transform(make):
    def wrapped(*args):
        outermost = not any(_is_t(a) for a in args)
        out = make()(*[_to_t(a) for a in args])
        return _to_np(out) if outermost else out
    return wrapped


def grad(fun, argnums=0):
    return _wrap_transform(lambda: _F.grad(_coerce_out(fun), argnums=argnums))


def hessian(fun, argnums=0):
    return _wrap_transform(lambda: _F.hessian(_coerce_out(fun), argnums=argnums))


def vmap(fun, in_axes=0, out_axes=0):
    return _wrap_transform(lambda: _F.vmap(_coerce_out(fun), in_dims=in_axes, out_dims=out_axes))


from . import numpy  # noqa: E402,F401

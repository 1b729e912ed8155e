"""`jax.numpy` stand-in: numpy when every argument is numpy, torch as soon as a torch tensor shows up."""
import numpy as _np
import torch as _torch

pi = _np.pi
newaxis = None
float64 = _np.float64


def _is_t(x):
    return isinstance(x, _torch.Tensor)


def _any_t(args):
    for a in args:
        if _is_t(a):
            return True
        if isinstance(a, (list, tuple)) and _any_t(a):
            return True
    return False


def _t(x):
    if _is_t(x):
        return x
    return _torch.as_tensor(_np.asarray(x, dtype=_np.float64))


class _At:
    def __init__(self, arr):
        self.arr = arr

    def __getitem__(self, idx):
        arr = self.arr

        class _Set:
            def set(self_inner, v):
                out = _np.array(arr, copy=True).view(Arr)
                out[idx] = _np.asarray(v)
                return out
        return _Set()


class Arr(_np.ndarray):
    """ndarray that (a) offers .at[idx].set(v), (b) defers to torch in mixed ndarray/Tensor arithmetic."""
    __array_priority__ = 1000

    @property
    def at(self):
        return _At(self)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if _any_t(inputs):
            name = ufunc.__name__
            tf = {'add': _torch.add, 'subtract': _torch.sub, 'multiply': _torch.mul, 'true_divide': _torch.div,
                  'divide': _torch.div, 'power': _torch.pow, 'negative': _torch.neg, 'exp': _torch.exp,
                  'sin': _torch.sin, 'cos': _torch.cos, 'sqrt': _torch.sqrt, 'log': _torch.log}[name]
            return tf(*[_t(i) for i in inputs])
        args = [_np.asarray(i) if isinstance(i, Arr) else i for i in inputs]
        out = getattr(ufunc, method)(*args, **kwargs)
        return out.view(Arr) if isinstance(out, _np.ndarray) else out

    def _bin(self, other, op, rop):
        if _is_t(other):
            return rop(_t(_np.asarray(self)), other)
        return op(other)

    def __add__(self, o): return self._bin(o, super().__add__, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, super().__radd__, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, super().__sub__, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, super().__rsub__, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, super().__mul__, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, super().__rmul__, lambda a, b: b * a)
    def __truediv__(self, o): return self._bin(o, super().__truediv__, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, super().__rtruediv__, lambda a, b: b / a)


def _v(x):
    return x.view(Arr) if isinstance(x, _np.ndarray) else x


def _unary(tf, nf):
    def f(x):
        return tf(x) if _is_t(x) else _v(nf(_np.asarray(x)))
    return f


exp = _unary(_torch.exp, _np.exp)
sin = _unary(_torch.sin, _np.sin)
cos = _unary(_torch.cos, _np.cos)
sqrt = _unary(_torch.sqrt, _np.sqrt)
log = _unary(_torch.log, _np.log)
abs = _unary(_torch.abs, _np.abs)
isnan = _unary(_torch.isnan, _np.isnan)
transpose = _unary(lambda x: x.t() if x.dim() == 2 else x, _np.transpose)


def array(x, dtype=None):
    return x if _is_t(x) else _v(_np.array(x, dtype=_np.float64 if dtype is None else dtype))


def asarray(x, dtype=None):
    return array(x, dtype)


def zeros(shape, dtype=None): return _v(_np.zeros(shape))
def ones(shape, dtype=None): return _v(_np.ones(shape))
def eye(n, dtype=None): return _v(_np.eye(n))
def arange(*a, **k): return _v(_np.arange(*a, **k))
def linspace(*a, **k): return _v(_np.linspace(*a, **k))
def meshgrid(*a, **k): return [_v(m) for m in _np.meshgrid(*[_np.asarray(x) for x in a], **k)]


def dot(a, b):
    return _torch.dot(_t(a), _t(b)) if _any_t((a, b)) else _np.dot(_np.asarray(a), _np.asarray(b))


def matmul(a, b):
    return _torch.matmul(_t(a), _t(b)) if _any_t((a, b)) else _v(_np.matmul(_np.asarray(a), _np.asarray(b)))


def trace(a):
    return _torch.trace(a) if _is_t(a) else _np.trace(_np.asarray(a))


def diag(a):
    return _torch.diag(a) if _is_t(a) else _v(_np.diag(_np.asarray(a)))


def sum(a, axis=None):
    if _is_t(a):
        return _torch.sum(a) if axis is None else _torch.sum(a, dim=axis)
    return _np.sum(_np.asarray(a), axis=axis)


def max(a):
    return _torch.max(a) if _is_t(a) else _np.max(_np.asarray(a))


def concatenate(seq, axis=0):
    if _any_t(seq):
        return _torch.cat([_t(s) for s in seq], dim=axis)
    return _v(_np.concatenate([_np.asarray(s) for s in seq], axis=axis))


def append(a, b, axis=None):
    if _any_t((a, b)):
        return _torch.cat([_t(a).reshape(-1), _t(b).reshape(-1)])
    return _v(_np.append(_np.asarray(a), _np.asarray(b), axis=axis))


def tile(a, reps):
    return _v(_np.tile(_np.asarray(a), reps))


def reshape(a, shape):
    return a.reshape(shape) if _is_t(a) else _v(_np.reshape(_np.asarray(a), shape))


from . import linalg  # noqa: E402,F401

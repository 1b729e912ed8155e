"""Process-wide device context for the host API (one handle = one GPU + one stream)."""
import os

import numpy as np

_ctx = None


def get_context():
    """Create (once) the gpk.Context on device $GPK_DEVICE (default: $LOCAL_RANK, else 0).
    Raises gpk.GpkError if libgpk.so is not built or no gfx950 device is visible -- there is no CPU fallback."""
    global _ctx
    if _ctx is None:
        import gpk
        dev = int(os.environ.get('GPK_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        _ctx = gpk.Context(dev)
    return _ctx


def eval_callback(fn, x1, x2):
    """Evaluate a user callback bdy(x1,x2)/rhs(x1,x2) on point columns.  The reference vmaps a JAX scalar function
    (src/PDEs.py:44-45); here numpy-vectorised callables are used directly and scalar-only callables (Python ints,
    comparisons, math.*) fall back to an element-wise loop -- only on the TypeError / ValueError such callables raise on arrays;
    every other exception of the callback propagates."""
    x1 = np.asarray(x1, dtype=np.float64); x2 = np.asarray(x2, dtype=np.float64)
    try:
        out = np.asarray(fn(x1, x2), dtype=np.float64)
        if out.shape == x1.shape:
            return out
        if out.ndim == 0:
            return np.full(x1.shape, float(out))
    except (TypeError, ValueError):
        # what a scalar-only callable raises when handed arrays: math.sin(array) -> TypeError ("only length-1 arrays can be converted"),
        # `if x1 > 0` -> ValueError ("truth value of an array is ambiguous"), float(array) -> TypeError.  Anything else -- a NameError
        # from a typo, a ZeroDivisionError, an IndexError -- is the user's bug and propagates from this first call, untouched.
        pass
    return np.array([float(fn(float(a), float(b))) for a, b in zip(x1, x2)], dtype=np.float64).reshape(x1.shape)

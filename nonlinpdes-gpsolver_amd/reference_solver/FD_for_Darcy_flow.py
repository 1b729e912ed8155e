"""Truth for -div(a grad u) = f on [0,1]^2 with u = 0 on the boundary (counterpart of the reference's
reference_solver/FD_for_Darcy_flow.py): flux-form 5-point finite differences, coefficient sampled at cell faces.
`fun_a(x1, x2)` and `f(x1, x2)` are numpy-vectorised (or scalar) callables; returns the (N+2, N+2) grid."""
import numpy as onp
import scipy.sparse as sparse
from scipy.sparse.linalg import spsolve


def _vec(fun, x1, x2):
    try:
        out = onp.asarray(fun(x1, x2), dtype=float)
        if out.shape == onp.broadcast(x1, x2).shape:
            return out
    except (TypeError, ValueError):                   # a scalar-only callback (math.sin, float(...)): evaluated point by point below;
        pass                                          # anything else the callback raises is the caller's bug and propagates
    return onp.vectorize(lambda a, b: float(fun(a, b)))(x1, x2)


def FD_Darcy_flow_2d(N, fun_a, f):
    h = 1.0 / (N + 1)
    face = (onp.arange(0, N + 1) + 0.5) * h           # cell faces
    node = onp.arange(1, N + 1) * h                   # interior nodes
    ax = _vec(fun_a, face[None, :] + 0 * node[:, None], node[:, None] + 0 * face[None, :])    # (N, N+1): faces in x1
    ay = _vec(fun_a, node[None, :] + 0 * face[:, None], face[:, None] + 0 * node[None, :])    # (N+1, N): faces in x2
    k = onp.arange(N * N).reshape(N, N)               # unknown index: row = x2 index, column = x1 index
    entries = [(k, k, ax[:, :-1] + ax[:, 1:] + ay[:-1, :] + ay[1:, :])]
    entries += [(k[:, :-1], k[:, 1:], -ax[:, 1:N]), (k[:, 1:], k[:, :-1], -ax[:, 1:N])]
    entries += [(k[:-1, :], k[1:, :], -ay[1:N, :]), (k[1:, :], k[:-1, :], -ay[1:N, :])]
    rows = onp.concatenate([e[0].ravel() for e in entries])
    cols = onp.concatenate([e[1].ravel() for e in entries])
    vals = onp.concatenate([onp.asarray(e[2], dtype=float).ravel() for e in entries])
    A = sparse.csc_matrix((vals / h ** 2, (rows, cols)), shape=(N * N, N * N))
    XX, YY = onp.meshgrid(node, node)
    u = spsolve(A, _vec(f, XX, YY).ravel())
    out = onp.zeros((N + 2, N + 2))
    out[1:N + 1, 1:N + 1] = u.reshape(N, N)
    return out

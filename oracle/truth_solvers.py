"""Classical truth generators used only to reproduce the reference's known-answer vectors.
*** TEST INFRASTRUCTURE ONLY *** (same rules as gp_oracle.py).

  fd_darcy_flow_2d   follows reference_solver/FD_for_Darcy_flow.py:8-33 (5-point flux-form FD, spsolve)
  cole_hopf_eikonal  follows reference_solver/Cole_Hopf_for_Eikonal.py:7-36
  burgers_cole_hopf  follows main_Burgers1d.py:87-92 (80-point Gauss-Hermite quadrature)
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _five_point(N, west_east, south_north):
    """Symmetric 5-point operator on an N x N interior grid, unknown k = i*N + j (i: x2 index, j: x1 index).
    west_east[i, j] (N, N+1) is the face coefficient between (i, j-1) and (i, j);
    south_north[i, j] (N+1, N) the one between (i-1, j) and (i, j)."""
    idx = np.arange(N * N).reshape(N, N)
    diag = west_east[:, :N] + west_east[:, 1:] + south_north[:N, :] + south_north[1:, :]
    rows = [idx.ravel()]; cols = [idx.ravel()]; vals = [diag.ravel()]
    # x1-neighbours
    rows += [idx[:, :-1].ravel(), idx[:, 1:].ravel()]
    cols += [idx[:, 1:].ravel(), idx[:, :-1].ravel()]
    c = -west_east[:, 1:N].ravel(); vals += [c, c]
    # x2-neighbours
    rows += [idx[:-1, :].ravel(), idx[1:, :].ravel()]
    cols += [idx[1:, :].ravel(), idx[:-1, :].ravel()]
    c = -south_north[1:N, :].ravel(); vals += [c, c]
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N * N, N * N))


def fd_darcy_flow_2d(N, fun_a, fun_f):
    """-div(a grad u) = f on [0,1]^2, u=0 on the boundary; returns the (N+2, N+2) grid incl. boundary zeros."""
    h = 1.0 / (N + 1)
    mid = (np.arange(0, N + 1) + 0.5) * h
    grid = np.arange(1, N + 1) * h
    we = fun_a(mid[None, :], grid[:, None]) * np.ones((N, N + 1))          # a at (x1 = mid_j, x2 = grid_i)
    sn = fun_a(grid[None, :], mid[:, None]) * np.ones((N + 1, N))          # a at (x1 = grid_j, x2 = mid_i)
    A = _five_point(N, we, sn) / (h * h)
    XX, YY = np.meshgrid(grid, grid)
    fv = fun_f(XX, YY) * np.ones((N, N))
    u = spla.spsolve(A.tocsc(), fv.ravel())
    out = np.zeros((N + 2, N + 2))
    out[1:N + 1, 1:N + 1] = u.reshape(N, N)
    return out


def cole_hopf_eikonal(N, epsilon):
    """|grad u|^2 = 1 + eps*Lap u, u=0 on the boundary, via v = exp(-u/eps): (I + eps^2 A/h^2) v = b."""
    h = 1.0 / (N + 1)
    grid = np.arange(1, N + 1) * h
    A = _five_point(N, np.ones((N, N + 1)), np.ones((N + 1, N)))
    b = np.zeros((N, N))
    s = epsilon ** 2 / h ** 2
    b[0, :] += s; b[N - 1, :] += s; b[:, 0] += s; b[:, N - 1] += s
    M = sp.identity(N * N, format='csr') + (epsilon ** 2) * A / (h ** 2)
    v = spla.spsolve(M.tocsc(), b.ravel())
    XX, YY = np.meshgrid(grid, grid)
    return XX, YY, (-epsilon * np.log(v)).reshape(N, N)


def burgers_cole_hopf(t, x, nu):
    """u(t,x) for u_t + u u_x = nu u_xx, u(0,x) = -sin(pi x)."""
    pts, wts = np.polynomial.hermite.hermgauss(80)
    t = np.asarray(t, float)[..., None]; x = np.asarray(x, float)[..., None]
    y = x - np.sqrt(4.0 * nu * t) * pts
    e = wts * np.exp(-np.cos(np.pi * y) / (2.0 * np.pi * nu))
    return -np.sum(np.sin(np.pi * y) * e, axis=-1) / np.sum(e, axis=-1)

"""Truth for the regularised Eikonal equation |grad u|^2 = 1 + eps*Lap u, u = 0 on the boundary of [0,1]^2
(counterpart of the reference's reference_solver/Cole_Hopf_for_Eikonal.py): with v = exp(-u/eps) the equation becomes
the linear problem (I - eps^2 Lap) v = 0 with v = 1 on the boundary, solved by 5-point finite differences."""
import numpy as onp
import scipy.sparse as sparse
from scipy.sparse.linalg import spsolve


def _laplacian_dirichlet(N):
    """minus the 5-point Laplacian times h^2 on an N x N interior grid (unknown k = i*N + j)"""
    one = sparse.diags([-onp.ones(N - 1), 2.0 * onp.ones(N), -onp.ones(N - 1)], [-1, 0, 1])
    eye = sparse.identity(N)
    return sparse.kron(eye, one) + sparse.kron(one, eye)


def solve_Eikonal(N, epsilon):
    h = 1.0 / (N + 1)
    pts = onp.arange(1, N + 1) * h
    XX, YY = onp.meshgrid(pts, pts)
    c = epsilon ** 2 / h ** 2
    rhs = onp.zeros((N, N))                         # boundary value v = 1 moved to the right-hand side
    rhs[0, :] += c; rhs[-1, :] += c; rhs[:, 0] += c; rhs[:, -1] += c
    M = (sparse.identity(N * N) + c * _laplacian_dirichlet(N)).tocsc()
    v = spsolve(M, rhs.ravel())
    return XX, YY, (-epsilon * onp.log(v)).reshape(N, N)

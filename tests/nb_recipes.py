"""RNG recipes of the reference notebooks (used by the known-answer tests).

The notebooks do NOT use src/sample_points.py: they draw the interior points as one (N,2) uniform block
(`Nonlinear_Elliptic_Equation.ipynb` cell 4, `Darcy_flow_IP_noisy.ipynb` cell 5,
`Regularized_Eikonal equation_eps1e-2.ipynb` cell 5) from numpy's legacy global RNG.
"""
import numpy as np


def notebook_sample_points(N_domain, N_boundary):
    X_domain = np.random.uniform(0.0, 1.0, (N_domain, 2))
    X_boundary = np.zeros((N_boundary, 2))
    q = int(N_boundary / 4)
    X_boundary[0:q, 0] = np.random.uniform(0.0, 1.0, q)              # bottom
    X_boundary[q:2 * q, 0] += 1                                      # right
    X_boundary[q:2 * q, 1] = np.random.uniform(0.0, 1.0, q)
    X_boundary[2 * q:3 * q, 0] = np.random.uniform(0.0, 1.0, q)      # top
    X_boundary[2 * q:3 * q, 1] += 1
    X_boundary[3 * q:4 * q, 1] = np.random.uniform(0.0, 1.0, q)      # left
    return X_domain, X_boundary


# stored outputs, copied digit for digit from the notebooks' `outputs` fields
ELLIPTIC_RATIO = 4394.531249999999
ELLIPTIC_J = [15114510.624764077, 10865.92700387972, 5525.78748299654, 5525.742194561648,
              5525.742194527177, 5525.742194527214]
ELLIPTIC_PTS_L2, ELLIPTIC_PTS_MAX = 0.0033197644840169582, 0.020552482929202115
ELLIPTIC_TEST_L2, ELLIPTIC_TEST_MAX = 0.00340243832517087, 0.02224968440427331

DARCY_RATIO_U = [19.999999999999996, 19.999999999999996, 3999.9999999999995]
DARCY_RATIO_A = [24.999999999999996, 24.999999999999996]
DARCY_J = [111470296.72405547, 28232.44456087123, 5806.836322823332, 639.6083504223275, 46.365270457253686,
           11.039350210349292, 10.543986043016446, 10.542958760850624, 10.542937737781788]

EIKONAL_RATIO = [20.661157024793386, 20.661157024793386, 4132.231404958678]
EIKONAL_J = [531208610.3530341, 59661343792278.38, 3709639434445.687, 227141010501.72964, 13103700672.978308,
             610675546.6356657, 14993577.597545214, 139767.26361026015, 67254.40814329864, 67146.13256088887,
             67184.4022631998]
EIKONAL_TEST_L2, EIKONAL_TEST_MAX = 0.02506445909677251, 0.06384742102752328


def darcy_a(x1, x2):
    s = np.sin(2 * np.pi * x1) + np.sin(2 * np.pi * x2)
    return np.exp(s) + np.exp(-s)

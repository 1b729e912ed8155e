"""matplotlib helpers of solver_GP (figures are cosmetic and outside the hot path; no LaTeX requirement)."""
import numpy as onp


def _plt():
    import matplotlib.pyplot as plt
    return plt


def scatter_points(eqn, with_data, title):
    plt = _plt()
    ax = plt.figure().add_subplot(111)
    series = [(eqn.X_domain, 'Interior nodes'), (eqn.X_boundary, 'Boundary nodes')]
    if with_data:
        series.append((eqn.X_domain[:eqn.N_data], 'Data nodes'))
    for X, label in series:
        ax.scatter(X[:, 0], X[:, 1], marker=None if with_data else 'x', label=label).set_clip_on(False)
    ax.legend(loc="upper right")
    plt.title(title)


def loss_history(eqn):
    plt = _plt()
    plt.figure()
    plt.plot(onp.arange(eqn.max_iter + 1), eqn.loss_hist)
    plt.yscale("log")
    plt.title('Loss function history')
    plt.xlabel('Gauss-Newton step')


def error_contour(XX, YY, err):
    plt = _plt()
    fig = plt.figure()
    cs = fig.add_subplot(111).contourf(XX, YY, err.reshape(XX.shape), 50, cmap=plt.cm.coolwarm)
    plt.xlabel('x_1')
    plt.ylabel('x_2')
    plt.title('Contour of errors')
    fig.colorbar(cs)
    plt.show()

"""CPU-side tests of the host layer (no GPU): API surface parity with the reference's src/ package, closed-form kernel
classes, bit-exact samplers, the C-ABI library's exported symbols, and loud failure without a device."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest

from oracle import gp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')


def test_kernel_classes_have_reference_method_names():
    from src.kernels import Anisotropic_Gaussian_kernel, Gaussian_kernel
    names19 = [n for n in O.KERNEL_METHODS if n != 'Delta_x_y_kappa']
    for n in names19:
        assert callable(getattr(Gaussian_kernel(), n)) and callable(getattr(Anisotropic_Gaussian_kernel(), n))
    assert callable(Anisotropic_Gaussian_kernel().Delta_x_y_kappa)
    sig = inspect.signature(Gaussian_kernel().Delta_x_Delta_y_kappa)
    assert list(sig.parameters) == ['x1', 'x2', 'y1', 'y2', 'sigma']


def test_kernel_classes_match_oracle():
    from src.kernels import Anisotropic_Gaussian_kernel, Gaussian_kernel
    rng = np.random.RandomState(0)
    x = rng.uniform(0, 1, (4, 200))
    for n in O.KERNEL_METHODS:
        if n != 'Delta_x_y_kappa':
            np.testing.assert_allclose(getattr(Gaussian_kernel(), n)(*x, 0.2), O.deriv_kernel(n, *x, 'Gaussian', 0.2), rtol=1e-13, atol=1e-12)
        np.testing.assert_allclose(getattr(Anisotropic_Gaussian_kernel(), n)(*x, [0.3, 0.5]),
                                   O.deriv_kernel(n, *x, 'anisotropic_Gaussian', [0.3, 0.5]), rtol=1e-13, atol=1e-12)
    # scalar in, scalar out like the reference
    assert np.ndim(Gaussian_kernel().kappa(0.1, 0.2, 0.3, 0.4, 0.2)) == 0


def test_samplers_bit_exact_against_reference_fixtures():
    from src import sample_points as SP
    d = np.load(os.path.join(G, 'sampling.npz'))
    for name in sorted({k.split('__')[0] for k in d.files}):
        nd, nb, a, b, c, e, td, seed = d[name + '__args']
        dom = np.array([[a, b], [c, e]])
        if name.startswith('grid'):
            Xd, Xb = SP.sampled_pts_grid(int(nd), int(nb), dom, bool(td))
        else:
            np.random.seed(int(seed))
            Xd, Xb = SP.sampled_pts_rdm(int(nd), int(nb), dom, bool(td))
            assert np.array_equal(np.random.uniform(0, 1, 3), d[name + '__tail'])      # same RNG consumption
        assert np.array_equal(Xd, d[name + '__Xd']) and np.array_equal(Xb, d[name + '__Xb']), name


@pytest.mark.reference
def test_samplers_against_live_reference_module():
    import importlib.util
    import sys
    from src import sample_points as SP
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location('ref_sp', '/root/reference/src/sample_points.py')
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    for nd, nb, td in [(17, 9, False), (23, 11, True), (900, 124, False)]:
        dom = np.array([[0, 1], [-1, 1]]) if td else np.array([[0, 1], [0, 1]])
        np.random.seed(5); a = ref.sampled_pts_rdm(nd, nb, dom, time_dependent=td)
        np.random.seed(5); b = SP.sampled_pts_rdm(nd, nb, dom, time_dependent=td)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        a = ref.sampled_pts_grid(nd, nb, dom, time_dependent=td); b = SP.sampled_pts_grid(nd, nb, dom, time_dependent=td)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_public_api_surface():
    """names the reference's main_*.py and solver.py touch"""
    from src import Gram_matrice, InverseProblems, PDEs, solver
    assert list(inspect.signature(Gram_matrice.Gram_matrix_assembly).parameters) == ['X_domain', 'X_boundary', 'eqn', 'kernel', 'kernel_parameter']
    assert list(inspect.signature(Gram_matrice.construct_Theta_test).parameters) == ['X_test', 'X_domain', 'X_boundary', 'eqn', 'kernel', 'kernel_parameter']
    common = ['get_bd', 'get_rhs', 'sampled_pts', 'get_sampled_points', 'Gram_matrix', 'Gram_Cholesky', 'loss', 'grad_loss',
              'Hessian_GN', 'GN_method', 'extend_sol']
    for cls in (PDEs.Nonlinear_elliptic2d, PDEs.Burgers, PDEs.Eikonal, InverseProblems.Darcy_flow2d):
        for m in common:
            assert callable(getattr(cls, m)), (cls.__name__, m)
    for m in ('GN_loss', 'loss_relaxed', 'grad_loss_relaxed', 'GN_loss_relaxed', 'Hessian_GN_relaxed', 'GN_relaxed_method'):
        assert callable(getattr(PDEs.Nonlinear_elliptic2d, m))
    assert list(inspect.signature(PDEs.Burgers.Hessian_GN).parameters) == ['self', 'z']          # ONE argument
    assert list(inspect.signature(PDEs.Eikonal.Hessian_GN).parameters) == ['self', 'z', 'z_old']
    assert inspect.signature(PDEs.Nonlinear_elliptic2d.__init__).parameters['m'].default == 3
    assert inspect.signature(PDEs.Burgers.__init__).parameters['nu'].default == 0.2
    assert inspect.signature(PDEs.Burgers.Gram_matrix).parameters['kernel_parameter'].default == [1 / 3, 1 / 20]
    assert inspect.signature(PDEs.Eikonal.__init__).parameters['eps'].default == 3
    assert inspect.signature(InverseProblems.Darcy_flow2d.Gram_matrix).parameters['nugget'].default == 1e-10
    for m in ('set_equation', 'get_sample', 'auto_sample', 'show_sample', 'get_sample_IP', 'auto_sample_IP', 'show_sample_IP',
              'get_observed_data', 'solve', 'show_loss_hist', 'collocation_pts_err', 'test', 'get_test_error', 'contour_of_test_err'):
        assert callable(getattr(solver.solver_GP, m)), m
    assert list(inspect.signature(solver.solver_GP.solve).parameters) == ['self', 'method', 'pen_lambda', 'print_option']


def test_callbacks_scalar_and_vectorised():
    from src._runtime import eval_callback
    x = np.linspace(0, 1, 7); y = np.linspace(-1, 1, 7)
    np.testing.assert_array_equal(eval_callback(lambda a, b: 0, x, y), np.zeros(7))                  # Python int
    np.testing.assert_allclose(eval_callback(lambda a, b: -np.sin(np.pi * b) * (a == 0) + 0 * (b == 0), x, y),
                               -np.sin(np.pi * y) * (x == 0))
    import math
    np.testing.assert_allclose(eval_callback(lambda a, b: math.sin(a) + b, x, y), np.sin(x) + y)      # scalar-only


def test_driver_flags_match_reference_defaults():
    """flag names / defaults of SURVEY 8b B2, read from the drivers' source (they run the solve at import time)"""
    pkg = os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')
    want = {
        'main_NonLinElliptic2d.py': ['--alpha', '--m', '--pen_lambda', "'Gaussian', 0.2, 1e-13, 900, 124", "'rdm', 4"],
        'main_Burgers1d.py': ['--alpha', '--nu', "default=0.02", "'anisotropic_Gaussian', [0.3, 0.05], 1e-5, 1000, 200", "'rdm', 8", '--randomseed", type=int, default=0'],
        'main_Eikonal2d.py': ['--eps", type=float, default=1e-1', "'Gaussian', 0.2, 1e-5, 1000, 200", "'zero', 8"],
        'main_DarcyFlow2d.py': ["'Gaussian', 0.2, 1e-8, 400, 100", '--N_data", type=int, default=60', '--noise_level", type=float, default=1e-3',
                                "'rdm', 8", '--randomseed", type=int, default=9999'],
    }
    for fn, needles in want.items():
        src = open(os.path.join(pkg, fn)).read()
        for n in needles:
            assert n in src, (fn, n)
    common = open(os.path.join(pkg, '_driver_common.py')).read()
    for flag in ('--kernel', '--kernel_parameter', '--nugget', '--nugget_type', '--sampled_type', '--N_domain', '--N_boundary',
                 '--method', '--initial_sol', '--GNsteps', '--step_size', '--print_hist', '--show_figure'):
        assert f'"{flag}"' in common, flag


# ---- the C-ABI library ---------------------------------------------------------------------------------------------
def _declared(headers):
    declared = set()
    for name in headers:
        hdr = open(os.path.join(ROOT, 'include', name)).read()
        hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)          # (prose in comments mentions calls like gpk_mg_rccl_init())
        declared |= set(re.findall(r'\b(gpk_[a-z0-9_]+)\s*\(', hdr))
    return declared - {'gpk_ctx'}


def test_library_exports_every_declared_symbol():
    """libgpk.so (the product) exports exactly the boundary -- gpk.h, gpk_mg.h -- plus the one per-handle tuning call of gpk_debug.h;
    libgpk_dev.so (tests / tools only) exports those and the development entry points of gpk_dev.h.  The product must NOT carry
    probes, micro-benchmarks or process-wide switches (round 4)."""
    import gpk
    product = _declared(('gpk.h', 'gpk_mg.h', 'gpk_debug.h'))
    dev_only = _declared(('gpk_dev.h',)) - product
    assert product and dev_only, 'no declarations parsed'
    for dev, declared in ((False, product), (True, product | dev_only)):
        path = gpk.library_path(dev=dev)
        if not os.path.exists(path):
            pytest.skip(f'{os.path.basename(path)} not built (run __graft_entry__.build())')
        lib = ctypes.CDLL(path)
        missing = [s for s in sorted(declared) if not hasattr(lib, s)]
        assert not missing, (path, missing)
        assert declared == set(gpk.declared_symbols(dev=dev)), declared ^ set(gpk.declared_symbols(dev=dev))
    lib = ctypes.CDLL(gpk.library_path())
    leaked = [s for s in sorted(dev_only) + ['gpk_debug_set'] if hasattr(lib, s)]
    assert not leaked, f'development entry points in the product library: {leaked}'


def test_gn_problem_struct_matches_header():
    """gpk/_lib.py mirrors `gpk_gn_problem` field by field (name, order, kind): a drift would shift every later field silently"""
    import gpk._lib as L
    hdr = open(os.path.join(ROOT, 'include', 'gpk.h')).read()
    body = re.search(r'typedef struct \{(.*?)\} gpk_gn_problem;', hdr, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r'(const\s+double\s*\*|int|double)\s*(.*)', decl)
        assert m, decl
        kind = {'int': ctypes.c_int, 'double': ctypes.c_double}.get(m.group(1), ctypes.c_void_p)
        for name in m.group(2).split(','):
            fields.append((name.strip().lstrip('*').strip(), kind))
    mirror = [(n, t) for n, t in L.GNProblemStruct._fields_]
    assert [n for n, _ in fields] == [n for n, _ in mirror]
    for (n, t), (_, u) in zip(fields, mirror):
        assert t is u, (n, t, u)


def test_no_cpu_fallback_without_device():
    """On a machine without a gfx950 device the product path must fail loudly (never route through the oracle)."""
    import gpk
    if not os.path.exists(gpk.library_path()):
        pytest.skip('libgpk.so not built')
    try:
        ctx = gpk.Context(0)
    except gpk.GpkError as e:
        assert 'no CPU fallback' in str(e)
        return
    ctx.close()          # a GPU is present: nothing to check here


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'nonlinpdes-gpsolver_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, os.path.join(dirpath, f)


def test_gn_worksize_query_runs_without_a_device():
    """gpk_gn_dims / gpk_gn_worksize are pure host functions (SURVEY 8b B3 "workspace queries"): sizes of the caller-owned arrays of
    the Gauss-Newton step for every system, the admissible leading dimension, the handle's own reservation."""
    import ctypes as C
    import gpk
    from gpk._lib import GNProblemStruct
    if not os.path.exists(gpk.library_path()):
        pytest.skip('libgpk.so not built')
    lib = gpk.load_library()
    cases = {0: (lambda Nd, Nb, Nk: (Nd, 2 * Nd + Nb)), 1: (lambda Nd, Nb, Nk: (3 * Nd, 4 * Nd + Nb)), 2: (lambda Nd, Nb, Nk: (3 * Nd, 4 * Nd + Nb)),
             3: (lambda Nd, Nb, Nk: (6 * Nd, 7 * Nd + Nb + Nk)), 4: (lambda Nd, Nb, Nk: (2 * Nd, 3 * Nd + Nb))}
    for system, dims in cases.items():
        for Nd, Nb, Nk in ((900, 124, 40), (37, 0, 5)):
            ps = GNProblemStruct()
            ps.system, ps.Nd, ps.Nb, ps.Ndata = system, Nd, Nb, Nk if system == 3 else 0
            nz, rows = C.c_int(), C.c_int()
            assert lib.gpk_gn_dims(C.byref(ps), C.byref(nz), C.byref(rows)) == 0
            assert (nz.value, rows.value) == dims(Nd, Nb, Nk)
            ld = C.c_int(); sb, hb, db, wb = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
            assert lib.gpk_gn_worksize(C.byref(ps), 0, C.byref(ld), C.byref(sb), C.byref(hb), C.byref(db), C.byref(wb)) == 0
            assert ld.value % 16 == 0 and nz.value + 1 <= ld.value < nz.value + 17
            assert sb.value == rows.value * ld.value * 8 and hb.value == (nz.value + 1) * ld.value * 8 and db.value == nz.value * 8
            assert wb.value == rows.value * 8                      # no inverted diagonal blocks supplied: only the vector of the exact in-step loss
            # with the factors and their inverted diagonal blocks supplied the handle also reserves the out-of-place solve buffer -- whatever
            # dinv_block says (0 means 256 to the step): the query mirrors what assemble_normal_equations decides (advisor, round 4)
            for blk in (0, 1024):
                ps.L, ps.ldl, ps.Dinv, ps.dinv_block = 4096, rows.value, 8192, blk          # (never dereferenced: host function)
                if system == 3:
                    ps.L2, ps.ldl2, ps.Dinv2 = 4096, rows.value, 8192
                assert lib.gpk_gn_worksize(C.byref(ps), 0, None, None, None, None, C.byref(wb)) == 0
                assert wb.value == rows.value * ld.value * 8 + rows.value * 8
            if system == 3:                                       # one factor without its blocks: the substitution schedule, nothing reserved
                ps.Dinv2 = None
                assert lib.gpk_gn_worksize(C.byref(ps), 0, None, None, None, None, C.byref(wb)) == 0 and wb.value == rows.value * 8
            ps.L = ps.Dinv = ps.L2 = ps.Dinv2 = None
            assert lib.gpk_gn_worksize(C.byref(ps), nz.value, None, None, None, None, None) < 0      # lds < nz + 1
    ps = GNProblemStruct(); ps.system = 99; ps.Nd = 10
    assert lib.gpk_gn_worksize(C.byref(ps), 0, None, None, None, None, None) < 0

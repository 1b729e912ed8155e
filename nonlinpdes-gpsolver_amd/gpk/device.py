"""Thin object layer over the C ABI: a context (handle + stream), device arrays with a padded leading dimension,
and the calls of include/gpk.h with numpy-friendly signatures."""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib as _libmod
from ._lib import GNProblemStruct, GpkError, load_library

LAYOUT = {'Nonlinear_elliptic': 0, 'Burgers': 1, 'Eikonal': 2, 'Darcy_u': 2, 'Darcy_a': 3}
KERNEL = {'Gaussian': 0, 'anisotropic_Gaussian': 1}
NUGGET = {'none': 0, 'identity': 1, 'adaptive': 2}
DINV_BLOCK = int(__import__('os').environ.get('GPK_DINV_BLOCK', '0'))   # rows per inverted diagonal block of a factor (256 .. 2048); 0 = by size


def dinv_block_for(n):
    """Rows per inverted diagonal block for a factor of order n: GPK_DINV_BLOCK when set, else 1024, or 2048 from order 6000 on (the
    solve phase then has 5 instead of 9 triangular products at BASELINE config 2: 3.3 -> 3.15 ms per step; iterates agree with the
    substitution path to ~1e-12 there, tests/test_gpu_parity.py::test_trsm_dinv and DESIGN.md section 4 'Numerics') -- but never more
    than a quarter of the order (round 3, advisor): multiplying by an explicit inverse is only conditionally stable, and a block that
    is half the matrix (1024 rows at BASELINE config 1, order 1924) made the iterates differ from the substitution path by 3e-10
    instead of 3e-13.  Orders up to 1024 get 256-row blocks."""
    if DINV_BLOCK:
        return DINV_BLOCK
    db = 2048 if n >= 6000 else 1024
    while db > 256 and db > n // 4:
        db //= 2
    return db
SYSTEM = {'Nonlinear_elliptic': 0, 'Burgers': 1, 'Eikonal': 2, 'Darcy_flow2d': 3, 'Nonlinear_elliptic_relaxed': 4}


def pad_ld(n, mult=16):
    """Leading dimension in elements: multiple of 16 doubles (128 B) so rows start on cache-line boundaries and the
    GEMM's 16-byte loads apply to every sub-block the recursion produces."""
    return ((int(n) + mult - 1) // mult) * mult


def kernel_params(kernel, kernel_parameter):
    if kernel == 'Gaussian':
        return (C.c_double * 2)(float(kernel_parameter), 0.0)
    if kernel == 'anisotropic_Gaussian':
        return (C.c_double * 2)(float(kernel_parameter[0]), float(kernel_parameter[1]))
    raise ValueError(f'unknown kernel {kernel!r}')


class DeviceArray:
    """Row-major float64 device buffer, (rows, cols) with leading dimension ld >= cols (vectors: cols == ld == 1)."""

    def __init__(self, ctx, rows, cols=1, ld=None, zero=False):
        self.ctx = ctx
        self.rows, self.cols = int(rows), int(cols)
        self.ld = int(ld) if ld is not None else (1 if self.cols == 1 else pad_ld(self.cols))
        self.nbytes = max(self.rows, 1) * self.ld * 8
        p = C.c_void_p()
        ctx._chk(ctx.lib.gpk_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value
        ctx._live.add(self)
        if zero:
            self.zero()

    def zero(self):
        self.ctx._chk(self.ctx.lib.gpk_memset(self.ctx.h, self.ptr, 0, self.nbytes))

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        a2 = a.reshape(self.rows, self.cols)
        self.ctx._chk(self.ctx.lib.gpk_memcpy2d_h2d(self.ctx.h, self.ptr, self.ld * 8, a2.ctypes.data, self.cols * 8,
                                                    self.cols * 8, self.rows))
        return self

    def download(self, rows=None, cols=None, row0=0, col0=0):
        rows = self.rows - row0 if rows is None else rows
        cols = self.cols - col0 if cols is None else cols
        out = np.empty((rows, cols), dtype=np.float64)
        if rows and cols:
            src = self.ptr + (row0 * self.ld + col0) * 8
            self.ctx._chk(self.ctx.lib.gpk_memcpy2d_d2h(self.ctx.h, out.ctypes.data, cols * 8, src, self.ld * 8, cols * 8, rows))
        return out[:, 0] if (self.cols == 1 and cols == 1) else out

    def clone(self):
        """device-to-device copy (same shape and leading dimension)"""
        out = DeviceArray(self.ctx, self.rows, self.cols, self.ld)
        self.ctx._chk(self.ctx.lib.gpk_memcpy_d2d(self.ctx.h, out.ptr, self.ptr, self.nbytes))
        return out

    def at(self, row=0, col=0):
        return self.ptr + (row * self.ld + col) * 8

    def free(self):
        """Release the buffer.  Context.close() frees every array still alive, so after close() this is a no-op."""
        ptr, self.ptr = getattr(self, 'ptr', None), None
        if ptr and self.ctx.h:
            self.ctx._live.discard(self)
            self.ctx._chk(self.ctx.lib.gpk_free(self.ctx.h, ptr))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GNProblem:
    """Device-side description of one equation's Gauss-Newton system (gpk_gn_problem)."""

    def __init__(self, ctx, system, Nd, Nb, rhs_f, bdy_g, L, p0=0.0, p1=0.0, pen_lambda=0.0, data_u=None, L2=None, dinv=True, structured=False,
                 cache_a=None):
        """structured: False (default: the reference's operation sequence every step), True / 1 (prepare_structured), 2 (+ prepare_gram).
        cache_a (Darcy only): keep the iteration-independent a-part of the step (prepare_darcy: bit-identical iterates, less work per step);
        None = on unless GPK_DARCY_CACHE=0.
        dinv: also compute the inverses of the diagonal blocks of the factor(s) once (gpk_trtri_diag; True = blocks of
        dinv_block_for(order) rows, or 256 / 512 / 1024 / 2048), so that the solve S = L^{-1}[A | F] of every step runs as GEMMs only."""
        self.ctx = ctx
        self.keep = []
        def dev(v):
            v = np.asarray(v, dtype=np.float64).ravel()
            d = DeviceArray(ctx, max(v.size, 1)).upload(v) if v.size else DeviceArray(ctx, 1)
            self.keep.append(d)
            return d
        self.rhs_f, self.bdy_g = dev(rhs_f), dev(bdy_g)
        self.data_u = dev(data_u) if data_u is not None else None
        self.L, self.L2 = L, L2
        s = GNProblemStruct()
        s.system = SYSTEM[system] if isinstance(system, str) else int(system)
        s.Nd, s.Nb = int(Nd), int(Nb)
        s.Ndata = 0 if data_u is None else int(np.asarray(data_u).size)
        s.p0, s.p1, s.pen_lambda = float(p0), float(p1), float(pen_lambda)
        s.rhs_f, s.bdy_g = self.rhs_f.ptr, self.bdy_g.ptr
        s.data_u = self.data_u.ptr if self.data_u is not None else None
        s.L, s.ldl = L.ptr, L.ld
        s.L2, s.ldl2 = (L2.ptr, L2.ld) if L2 is not None else (None, 0)
        block = dinv_block_for(L.rows) if dinv is True else int(dinv)
        self.Dinv = ctx.trtri_diag(L, block=block) if dinv else None
        self.Dinv2 = ctx.trtri_diag(L2, block=block) if (dinv and L2 is not None) else None
        s.Dinv = self.Dinv.ptr if self.Dinv is not None else None
        s.Dinv2 = self.Dinv2.ptr if self.Dinv2 is not None else None
        s.dinv_block = block if dinv else 0
        self.struct = s
        nz, rows = C.c_int(), C.c_int()
        ctx._chk(ctx.lib.gpk_gn_dims(C.byref(s), C.byref(nz), C.byref(rows)))
        self.nz, self.rows = nz.value, rows.value
        self._S = self._H = self._delta = self._work = None
        self.W1 = self.W2 = self.v0 = self.G = self.pvec = None
        self.Wa = self.Ha = None
        self.darcy_prepare_ms = self.structured_prepare_ms = None
        if cache_a is None:
            cache_a = os.environ.get('GPK_DARCY_CACHE', '1') != '0'
        if cache_a and s.system == SYSTEM['Darcy_flow2d'] and dinv:
            self.prepare_darcy()
        if structured:
            self.prepare_structured()
            if int(structured) == 2:
                self.prepare_gram()

    def prepare_structured(self):
        """OPTIONAL: precompute the z-independent solves once (gpk_gn_structured_prepare).  Elliptic system: W1 = L^{-1}[I;0;0],
        W2 = L^{-1}[0;I;0], v0 = L^{-1}F(0); every later gn_step forms [L^{-1}A(z) | L^{-1}F(z)] = [W1 diag(d) + W2 | v0 + W1 a + W2 z]
        in one memory-bound pass instead of the triangular solve.  Burgers / Eikonal / Darcy (round 6): A(z) = A1 diag(d(z)) + A2 with
        constant 0/1 patterns, W1 = L^{-1}A1, W2 = L^{-1}A2; the step forms L^{-1}A(z) = W1 diag(d(z)) + W2 and solves only the F column.
        Not the reference's per-step operation sequence: opt-in."""
        import time
        if self.struct.system == SYSTEM['Nonlinear_elliptic_relaxed']:
            raise GpkError('structured solve: not available for the relaxed system')
        self.ctx.synchronize(); t0 = time.perf_counter()
        S, _, _, _ = self.workspace()
        ld = S.ld
        self.W1 = DeviceArray(self.ctx, self.rows, self.nz + 1, ld)
        self.W2 = DeviceArray(self.ctx, self.rows, self.nz + 1, ld)
        self.v0 = DeviceArray(self.ctx, self.rows)
        self.ctx._chk(self.ctx.lib.gpk_gn_structured_prepare(self.ctx.h, C.byref(self.struct), S.ptr, S.ld, self.W1.ptr, self.W2.ptr,
                                                             self.v0.ptr, ld))
        self.struct.W1, self.struct.W2, self.struct.v0, self.struct.ldw = self.W1.ptr, self.W2.ptr, self.v0.ptr, ld
        self.ctx.synchronize(); self.structured_prepare_ms = 1e3 * (time.perf_counter() - t0)

    def prepare_darcy(self):
        """Darcy: the a-part rows of GN_loss ([w1; w2; w0] against L_a, reference src/InverseProblems.py:137-146) do not involve z_old, so
        W_a = L_a^{-1} A_a and W_a^T W_a are the same in every step: computed once here with the step's own launches (gpk_gn_darcy_prepare),
        after which every gn_step skips that solve and that product.  Bit-identical iterates; 2 x (3 N_d)^2 doubles."""
        import time
        S, _, _, _ = self.workspace()
        na = 3 * self.struct.Nd
        ld = pad_ld(na)
        self.Wa = DeviceArray(self.ctx, na, na, ld)
        self.Ha = DeviceArray(self.ctx, na, na, ld)
        self.ctx.synchronize(); t0 = time.perf_counter()
        self.ctx._chk(self.ctx.lib.gpk_gn_darcy_prepare(self.ctx.h, C.byref(self.struct), S.ptr, S.ld, self.Wa.ptr, ld, self.Ha.ptr, ld))
        self.ctx.synchronize(); self.darcy_prepare_ms = 1e3 * (time.perf_counter() - t0)
        self.struct.Wa, self.struct.ldwa, self.struct.Ha, self.struct.ldha = self.Wa.ptr, ld, self.Ha.ptr, ld

    def prepare_gram(self):
        """OPTIONAL second level (needs prepare_structured): the Gram blocks G = [W1 W2]^T [W1 W2] and W^T v0 once
        (gpk_gn_gram_prepare); every later gn_step assembles the bordered matrix in O(nz^2) -- no solve, no product."""
        ldg = pad_ld(self.nz)
        self.G = DeviceArray(self.ctx, 4 * self.nz, self.nz, ldg)
        self.pvec = DeviceArray(self.ctx, 2 * self.nz + 1)
        self.ctx._chk(self.ctx.lib.gpk_gn_gram_prepare(self.ctx.h, C.byref(self.struct), self.G.ptr, ldg, self.pvec.ptr))
        self.struct.G, self.struct.ldg, self.struct.pvec = self.G.ptr, ldg, self.pvec.ptr

    def workspace(self):
        if self._S is None:
            ld = pad_ld(self.nz + 1)
            self._S = DeviceArray(self.ctx, self.rows, self.nz + 1, ld)
            self._H = DeviceArray(self.ctx, self.nz + 1, self.nz + 1, ld)
            self._delta = DeviceArray(self.ctx, self.nz)
            self._work = DeviceArray(self.ctx, self.rows)
        return self._S, self._H, self._delta, self._work

    def release_workspace(self):
        for a in (self._S, self._H, self._delta, self._work):
            if a is not None:
                a.free()
        self._S = self._H = self._delta = self._work = None

    def free(self):
        """Release everything this problem allocated (workspace, inverted diagonal blocks, the prepared operators of the structured modes,
        the cached Darcy a-part, its copies of the right-hand sides); the factors L / L2 belong to the caller.  The struct is inert afterwards."""
        self.release_workspace()
        for name in ('Dinv', 'Dinv2', 'W1', 'W2', 'v0', 'G', 'pvec', 'Wa', 'Ha'):
            a = getattr(self, name, None)
            if a is not None:
                a.free()
            setattr(self, name, None)
        for a in self.keep:
            a.free()
        self.keep = []
        s = self.struct
        s.Dinv = s.Dinv2 = s.W1 = s.W2 = s.v0 = s.G = s.pvec = s.Wa = s.Ha = None
        s.rhs_f = s.bdy_g = s.data_u = None


class Context:
    """One handle = one device + one stream.  Raises GpkError if the library or the device is missing."""

    def __init__(self, device=0, dev=False):
        """dev=True: the handle lives in libgpk_dev.so, the development build with the superseded kernel variants, probes and
        micro-benchmarks (include/gpk_dev.h) -- tests and tools only"""
        self.dev = bool(dev)
        self.lib = load_library(dev=self.dev)
        h = C.c_void_p()
        rc = self.lib.gpk_create(int(device), C.byref(h))
        if rc != 0:
            raise GpkError(f'gpk_create(device={device}) failed with {rc}: no usable gfx950 device '
                           '(this library has no CPU fallback)')
        self.h = h
        self._live = weakref.WeakSet()          # device arrays allocated through this context and not yet freed
        self.tune_rejected = {}
        for k, v in _libmod.TUNE_DEFAULTS.items():        # development switches in force for this process (GPK_DEBUG_SET / gpk.debug_set)
            if self.lib.gpk_tune(self.h, int(k), int(v)) != 0:
                # (a value that selects a superseded design while this handle lives in the product library, which does not contain it)
                self.tune_rejected[int(k)] = int(v)
                import warnings
                warnings.warn(f'gpk: tuning key {k} = {v} is not available in {"libgpk_dev.so" if self.dev else "libgpk.so"}; this handle keeps its default')
        _libmod.LIVE_CONTEXTS.add(self)

    def tune(self, key, value):
        """per-handle development / tuning switch (gpk_tune; keys: GpkTune in csrc/gpk_common.h, tools/README.md)"""
        self._chk(self.lib.gpk_tune(self.h, int(key), int(value)))

    def _chk(self, rc):
        if rc < 0:
            raise GpkError(f'libgpk error {rc}: {self.lib.gpk_last_error(self.h).decode()}')
        return rc

    def close(self):
        """Free every device array still alive (their handles become inert), then destroy the handle."""
        if getattr(self, 'h', None):
            for a in list(self._live):
                a.free()
            self.lib.gpk_destroy(self.h)
            self.h = None

    def synchronize(self):
        self._chk(self.lib.gpk_synchronize(self.h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, clk, hbm = C.c_int(), C.c_int(), C.c_size_t()
        self._chk(self.lib.gpk_device_info(self.h, name, 256, C.byref(cus), C.byref(hbm), C.byref(clk)))
        return dict(name=name.value.decode(), compute_units=cus.value, hbm_bytes=hbm.value, clock_khz=clk.value)

    # ---- arrays ----
    def empty(self, rows, cols=1, ld=None):
        return DeviceArray(self, rows, cols, ld)

    def array(self, a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 1:
            return DeviceArray(self, a.size).upload(a)
        return DeviceArray(self, a.shape[0], a.shape[1]).upload(a)

    def points(self, X):
        """(n,2) point set, contiguous (ld = 2) as the C ABI expects"""
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 2)
        return DeviceArray(self, max(X.shape[0], 1), 2, ld=2).upload(X) if X.shape[0] else DeviceArray(self, 1, 2, ld=2)

    def timer_start(self):
        self._chk(self.lib.gpk_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double()
        self._chk(self.lib.gpk_timer_stop(self.h, C.byref(ms)))
        return ms.value

    # ---- assembly ----
    def assemble(self, layout, kernel, kernel_parameter, Xd, Xb, nugget=0.0, nugget_type='none', out=None):
        Xd = np.ascontiguousarray(Xd, dtype=np.float64); Xb = np.ascontiguousarray(Xb, dtype=np.float64).reshape(-1, 2)
        Nd, Nb = Xd.shape[0], Xb.shape[0]
        lay = LAYOUT[layout]
        N = {0: 2 * Nd + Nb, 1: 4 * Nd + Nb, 2: 4 * Nd + Nb, 3: 3 * Nd}[lay]
        dXd, dXb = self.points(Xd), self.points(Xb)
        T = out if out is not None else DeviceArray(self, N, N)
        ratios = (C.c_double * 3)()
        self._chk(self.lib.gpk_assemble(self.h, lay, KERNEL[kernel], kernel_params(kernel, kernel_parameter),
                                        dXd.ptr, Nd, dXb.ptr, Nb, float(nugget), NUGGET[nugget_type], T.ptr, T.ld, ratios))
        self.synchronize()
        return T, list(ratios)

    def assemble_test(self, layout, kernel, kernel_parameter, Xt, Xd, Xb):
        Xt = np.ascontiguousarray(Xt, dtype=np.float64); Xd = np.ascontiguousarray(Xd, dtype=np.float64)
        Xb = np.ascontiguousarray(Xb, dtype=np.float64).reshape(-1, 2)
        Nt, Nd, Nb = Xt.shape[0], Xd.shape[0], Xb.shape[0]
        lay = LAYOUT[layout]
        N = {0: 2 * Nd + Nb, 1: 4 * Nd + Nb, 2: 4 * Nd + Nb, 3: 3 * Nd}[lay]
        dXt, dXd, dXb = self.points(Xt), self.points(Xd), self.points(Xb)
        out = DeviceArray(self, Nt, N)
        self._chk(self.lib.gpk_assemble_test(self.h, lay, KERNEL[kernel], kernel_params(kernel, kernel_parameter),
                                             dXt.ptr, Nt, dXd.ptr, Nd, dXb.ptr, Nb, out.ptr, out.ld))
        self.synchronize()
        return out

    def extend(self, layout, kernel, kernel_parameter, Xt, Xd, Xb, coeff):
        Xt = np.ascontiguousarray(Xt, dtype=np.float64); Xd = np.ascontiguousarray(Xd, dtype=np.float64)
        Xb = np.ascontiguousarray(Xb, dtype=np.float64).reshape(-1, 2)
        Nt, Nd, Nb = Xt.shape[0], Xd.shape[0], Xb.shape[0]
        dXt, dXd, dXb = self.points(Xt), self.points(Xd), self.points(Xb)
        dc = coeff if isinstance(coeff, DeviceArray) else self.array(coeff)
        out = DeviceArray(self, Nt)
        self._chk(self.lib.gpk_extend(self.h, LAYOUT[layout], KERNEL[kernel], kernel_params(kernel, kernel_parameter),
                                      dXt.ptr, Nt, dXd.ptr, Nd, dXb.ptr, Nb, dc.ptr, out.ptr))
        self.synchronize()
        return out

    def error_metrics(self, truth, approx):
        """(|truth - approx| as a numpy array, its maximum, sqrt(sum of squares / n)) computed on the device (gpk_error_metrics);
        truth / approx: host arrays or DeviceArrays of the same length"""
        dt = truth if isinstance(truth, DeviceArray) else self.array(np.asarray(truth, dtype=np.float64).ravel())
        da = approx if isinstance(approx, DeviceArray) else self.array(np.asarray(approx, dtype=np.float64).ravel())
        n = dt.rows
        if da.rows != n:
            raise ValueError(f'error_metrics: {n} truth values against {da.rows} approximations')
        err = DeviceArray(self, n)
        mx, l2 = C.c_double(), C.c_double()
        self._chk(self.lib.gpk_error_metrics(self.h, n, dt.ptr, da.ptr, err.ptr, C.byref(mx), C.byref(l2)))
        return err.download(), mx.value, l2.value

    # ---- dense ----
    def potrf(self, A, n=None):
        n = A.rows if n is None else n
        info = C.c_int()
        self._chk(self.lib.gpk_potrf(self.h, A.ptr, n, A.ld, C.byref(info)))
        return self._chk_info(info.value)

    @staticmethod
    def _chk_info(info):
        """> 0 is LAPACK's info (first non-positive pivot); < 0 means a bounded device-side wait expired (lost launch)."""
        if info < 0:
            raise GpkError(f'libgpk: a device-side wait expired inside the factorisation (info = {info}); the result is invalid')
        return info

    def tril(self, A, n=None):
        self._chk(self.lib.gpk_tril(self.h, A.ptr, A.rows if n is None else n, A.ld))

    def symmetrize(self, A, n=None):
        self._chk(self.lib.gpk_symmetrize_lower(self.h, A.ptr, A.rows if n is None else n, A.ld))

    def trsm(self, L, B, trans=False, n=None, nrhs=None):
        n = L.rows if n is None else n
        nrhs = B.cols if nrhs is None else nrhs
        self._chk(self.lib.gpk_trsm(self.h, int(trans), L.ptr, n, L.ld, B.ptr, nrhs, B.ld))

    def trtri_diag(self, L, n=None, block=None):
        """inverses of the block x block diagonal blocks of the factor L -> (n, block) device array (gpk_trtri_diag)"""
        n = L.rows if n is None else n
        block = dinv_block_for(n) if block is None else int(block)
        D = DeviceArray(self, n, block, ld=block)
        self._chk(self.lib.gpk_trtri_diag(self.h, L.ptr, n, L.ld, D.ptr, block))
        return D

    def trsm_dinv(self, L, Dinv, B, X, n=None, nrhs=None, lead=0):
        """X <- L^{-1} B through the inverted diagonal blocks (B becomes scratch; X zero on entry when lead > 0)"""
        n = L.rows if n is None else n
        nrhs = B.cols if nrhs is None else nrhs
        self._chk(self.lib.gpk_trsm_dinv(self.h, L.ptr, Dinv.ptr, Dinv.ld, n, L.ld, B.ptr, nrhs, B.ld, X.ptr, X.ld, int(lead)))

    def potrs(self, L, B, n=None, nrhs=None):
        n = L.rows if n is None else n
        nrhs = B.cols if nrhs is None else nrhs
        self._chk(self.lib.gpk_potrs(self.h, L.ptr, n, L.ld, B.ptr, nrhs, B.ld))

    def gemm(self, ta, tb, m, n, k, alpha, A, B, beta, Cm):
        self._chk(self.lib.gpk_gemm(self.h, int(ta), int(tb), m, n, k, float(alpha), A.ptr, A.ld, B.ptr, B.ld, float(beta), Cm.ptr, Cm.ld))

    def syrk(self, n, k, alpha, A, beta, Cm, full=False):
        self._chk(self.lib.gpk_syrk(self.h, n, k, float(alpha), A.ptr, A.ld, float(beta), Cm.ptr, Cm.ld, int(full)))

    # ---- Gauss-Newton ----
    def gn_step(self, prob, z, step_size=1.0):
        S, H, delta, _ = prob.workspace()
        loss, info = C.c_double(), C.c_int()
        self._chk(self.lib.gpk_gn_step(self.h, C.byref(prob.struct), z.ptr, float(step_size), S.ptr, S.ld, H.ptr, H.ld,
                                       delta.ptr, C.byref(loss), C.byref(info)))
        return loss.value, self._chk_info(info.value)

    def gn_loss(self, prob, z):
        _, _, _, work = prob.workspace()
        loss = C.c_double()
        self._chk(self.lib.gpk_gn_loss(self.h, C.byref(prob.struct), z.ptr, work.ptr, C.byref(loss)))
        return loss.value

    def gn_hessian_grad(self, prob, z):
        S, H, delta, _ = prob.workspace()
        self._chk(self.lib.gpk_gn_hessian_grad(self.h, C.byref(prob.struct), z.ptr, S.ptr, S.ld, H.ptr, H.ld, delta.ptr))
        self.synchronize()
        return H.download(prob.nz, prob.nz), delta.download()

    def gn_measurement(self, prob, z):
        _, _, _, work = prob.workspace()
        self._chk(self.lib.gpk_gn_measurement(self.h, C.byref(prob.struct), z.ptr, work.ptr))
        self.synchronize()
        return work.download()

    # ---- per-phase timing of gn_step ----
    def prof_enable(self, on=True):
        self._chk(self.lib.gpk_prof_enable(self.h, int(on)))

    def prof_read(self):
        ms = (C.c_double * 4)()
        n = C.c_int()
        self._chk(self.lib.gpk_prof_read(self.h, ms, C.byref(n)))
        pipelined, cus, syrk_launch = C.c_int(), C.c_int(), C.c_double()
        self._chk(self.lib.gpk_prof_read_pipeline(self.h, C.byref(pipelined), C.byref(syrk_launch), C.byref(cus)))
        # pipelined: syrk_ms is the wall time of the fused product + factorisation phase and potrf_ms is 0;
        # syrk_launch_ms = the SYRK launches themselves (events on the stream they ran on)
        # flops the matrix-product launches of those steps executed, counted by the launch logic itself (gpk_prof_read_flops)
        fl, nl = (C.c_double * 4)(), (C.c_long * 4)()
        self._chk(self.lib.gpk_prof_read_flops(self.h, fl, nl))
        return dict(steps=n.value, trsm_ms=ms[0], syrk_ms=ms[1], potrf_ms=ms[2], trsv_update_ms=ms[3],
                    pipelined=bool(pipelined.value), syrk_launch_ms=syrk_launch.value, chain_cus=cus.value,
                    solve_flops=fl[0], potrf_update_flops=fl[1], product_flops=fl[2],
                    solve_launches=nl[0], potrf_update_launches=nl[1], product_launches=nl[2])

    def prof_read_assembly(self):
        """milliseconds of the evaluator launch of the last assemble call issued with prof_enable(True)"""
        ms = C.c_double()
        self._chk(self.lib.gpk_prof_read_assembly(self.h, C.byref(ms)))
        return ms.value

    # ---- micro-benchmarks (development build only: Context(dev=True)) ----
    def ubench_mfma_f64(self, iters=20000):
        v = C.c_double()
        self._chk(self.lib.gpk_ubench_mfma_f64(self.h, iters, C.byref(v)))
        return v.value

    def ubench_latency(self, mode):
        v = C.c_double()
        self._chk(self.lib.gpk_ubench_latency(self.h, int(mode), C.byref(v)))
        return v.value

    def ubench_hbm_write(self, nbytes=1 << 30, iters=10):
        v = C.c_double()
        self._chk(self.lib.gpk_ubench_hbm_write(self.h, nbytes, iters, C.byref(v)))
        return v.value

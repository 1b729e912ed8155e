import sys; sys.path.insert(0, "nonlinpdes-gpsolver_amd")
import ctypes as C, numpy as np, gpk
ctx = gpk.Context(0, dev=True)
for mode in (0, 1):
    n = 4096
    out = (C.c_int * n)()
    ctx._chk(ctx.lib.gpk_ubench_xcc_map(ctx.h, n, mode, out))
    a = np.array(list(out))
    print('mode', mode, 'first 24:', a[:24].tolist(), ' match b%8:', float(np.mean(a == (np.arange(n) % 8))), 'counts', np.bincount(a, minlength=8).tolist())

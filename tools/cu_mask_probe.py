#!/usr/bin/env python3
"""Which CUs does a CU-masked stream use?  (bit i of the mask -> which XCD / CU)  Development probe."""
import sys; sys.path.insert(0, "nonlinpdes-gpsolver_amd")
import ctypes as C, numpy as np, gpk
ctx = gpk.Context(0, dev=True)
n = 4096
for first, nbits in ((0, 8), (0, 32), (0, 64), (32, 32), (64, 192), (0, 256), (8, 8), (0, 1), (1, 1), (8, 1)):
    out = (C.c_int * n)()
    ctx._chk(ctx.lib.gpk_ubench_cu_census(ctx.h, first, nbits, n, out))
    a = np.array(list(out))
    xcc = a & 0xff; hw = a >> 8
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    print(f'bits [{first},{first + nbits}): distinct (xcc,se,sh,cu) = {len(ids)}; per XCC: {np.bincount(xcc, minlength=8).tolist()}',
          sorted(ids)[:6] if nbits <= 8 else '')

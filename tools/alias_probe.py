#!/usr/bin/env python3
"""What happens to the two-partition pipeline when the process already owns other streams (torch's stream pool): runs Gauss-Newton
steps at n_z = 1100 on a fresh context after creating torch streams, reports whether the pipeline stayed on, the handle's message
and the step time.  Usage: alias_probe.py [n_torch_streams [n_raw_hip_streams]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, 'nonlinpdes-gpsolver_amd')
sys.path.insert(0, '.')
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
streams = [torch.cuda.Stream() for _ in range(n)]
for s in streams:
    with torch.cuda.stream(s):
        torch.zeros(16, device='cuda').sum().item()
import gpk
from oracle import gp_oracle as O
ctx = gpk.Context(0)
nraw = int(sys.argv[2]) if len(sys.argv) > 2 else 0                 # raw HIP streams created AFTER the context, BEFORE the first step
if nraw:
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    raw = [ctypes.c_void_p() for _ in range(nraw)]
    for r in raw:
        assert hip.hipStreamCreateWithFlags(ctypes.byref(r), 1) == 0
rng = np.random.RandomState(3)
Nd, Nb = int(os.environ.get("PROBE_ND", "2000")), 200
Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
f = O.elliptic_rhs(Xd[:, 0], Xd[:, 1]); g = O.elliptic_truth(Xb[:, 0], Xb[:, 1])
T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-10, 'adaptive')
assert ctx.potrf(T) == 0
prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
z = ctx.array(rng.normal(size=Nd))
ctx.prof_enable(True) if hasattr(ctx, 'prof_enable') else None
for it in range(6):
    t0 = time.perf_counter()
    loss, info = ctx.gn_step(prob, z, 1.0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f'step {it}: loss {loss:.6e} info {info} {dt*1e3:.2f} ms')
truth = O.elliptic_truth(Xd[:, 0], Xd[:, 1])
print('rms err', np.sqrt(np.mean((z.download().ravel() - truth) ** 2)))
print('handle message:', ctx.lib.gpk_last_error(ctx.h))
try:
    print('prof', ctx.prof_read())
except Exception as e:
    print('prof_read:', e)

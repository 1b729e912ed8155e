"""A/B of the exact in-step loss (gpk_tune key 52) at BASELINE config 2 (and any workload of bench.WORKLOADS): ms per gpk_gn_step with
0 = approximate free number (rounds 2-4), 1 = true substitution in front of the solve (default); and gpk_gn_step + gpk_gn_loss (round 4's
product sequence).  Prints phases from the library's events.  (Round 5 also measured the chain overlapped with the solve phase on the chain
partition, with and without moving the first solve launches to the GEMM partition: 7.39 / 7.24 ms per step against 6.97 serial at config 2.)
    python tools/exact_loss_ab.py [c2|n10k|c1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import numpy as np
import bench, gpk

wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
Nd, Nb, _, desc = bench.WORKLOADS[wl]
ctx = gpk.Context(0)
Xd, Xb, f, g, z0 = bench.synthetic_problem(Nd, Nb)
T, _ = ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-13, 'adaptive')
assert ctx.potrf(T) == 0
prob = gpk.GNProblem(ctx, 'Nonlinear_elliptic', Nd, Nb, f, g, T, p0=1.0, p1=3.0)
prob.workspace()
for label, key, sep in (('approximate (52=0)', 0, False), ('exact, chain next to the END of the step (52=1, default)', 1, False),
                        ('exact, in front of the solve (52=2)', 2, False), ('approximate + gpk_gn_loss call', 0, True), ('exact, default again', 1, False)):
    ctx.tune(52, key)
    z = ctx.array(z0)
    for _ in range(3):
        ctx.gn_step(prob, z)
    ctx.prof_enable(True)
    ctx.synchronize(); t0 = time.perf_counter()
    n = 20
    losses = []
    for _ in range(n):
        l = ctx.gn_step(prob, z)[0]
        if sep:
            l = ctx.gn_loss(prob, z)
        losses.append(l)
    ctx.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    pr = ctx.prof_read(); ctx.prof_enable(False)
    k = max(pr['steps'], 1)
    print(f'{wl} {label:58s}: {dt:7.3f} ms/step  solve {pr["trsm_ms"]/k:.3f}  product+potrf {pr["syrk_ms"]/k:.3f}  tail {pr["trsv_update_ms"]/k:.3f}  loss[-1] {losses[-1]:.12e}', flush=True)
    z.free()
ctx.close()

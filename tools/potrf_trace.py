#!/usr/bin/env python3
"""One gpk_potrf of order N in a given mode, for a kernel trace: python tools/potrf_trace.py N mode   (mode: seq | la32 | la64 | ll64)
and, with `summary <dir>`, the per-queue timeline of the LAST factorisation of such a trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == 'summary':
    import csv, glob
    f = max(glob.glob(os.path.join(sys.argv[2], '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    asm = [i for i, r in enumerate(rows) if 'assemble' in r['Kernel_Name']]
    rows = rows[asm[-1] + 1:]                                     # everything after the last assembly = the last factorisation
    t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
    print(f'factorisation: {len(rows)} kernels, {(t1 - t0) / 1e3:.1f} us from first start to last end')
    for q in sorted({r['Queue_Id'] for r in rows}):
        rq = [r for r in rows if r['Queue_Id'] == q]
        pan = [r for r in rq if 'potrf_panel' in r['Kernel_Name']]
        gem = [r for r in rq if 'gemm_' in r['Kernel_Name']]
        d = lambda rs: sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e3
        span = (max(int(r['End_Timestamp']) for r in rq) - min(int(r['Start_Timestamp']) for r in rq)) / 1e3
        print(f'  queue {q}: {len(rq)} kernels, busy {d(rq):8.1f} us of a {span:8.1f} us span; {len(pan)} panel kernels {d(pan):8.1f} us'
              + (f' (avg {d(pan) / len(pan):.1f}, max {max((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in pan):.1f})' if pan else '')
              + f'; {len(gem)} product launches {d(gem):8.1f} us')
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import numpy as np
import gpk
from src.sample_points import sampled_pts_rdm
N, mode = int(sys.argv[1]), sys.argv[2]
ctx = gpk.Context(0, dev=True)
for k, v in {'seq': {20: 0}, 'la32': {20: 10 ** 6, 54: 1, 13: 32}, 'la64': {20: 10 ** 6, 54: 1, 13: 64}, 'll64': {20: 10 ** 6, 54: 0, 13: 64}}[mode].items():
    ctx.tune(k, v)
Nd = N * 10 // 21; Nb = N - 2 * Nd
np.random.seed(0)
Xd, Xb = sampled_pts_rdm(Nd, Nb, np.array([[0, 1], [0, 1]]))
T = ctx.empty(2 * Xd.shape[0] + Xb.shape[0], 2 * Xd.shape[0] + Xb.shape[0])
for rep in range(3):
    ctx.assemble('Nonlinear_elliptic', 'Gaussian', 0.2, Xd, Xb, 1e-10, 'adaptive', out=T)
    ctx.timer_start(); info = ctx.potrf(T); ms = ctx.timer_stop()
print(f'N={T.rows} mode {mode}: {ms:.2f} ms, info {info}')

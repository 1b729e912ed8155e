#!/usr/bin/env python3
"""Round 6, verdict item 6: store policy of the Gram evaluator (assemble2_kernel, csrc/gpk_assemble.hip).  gpk_tune key 55:
0 = plain global_store_dwordx4 (default cache policy), 1 = non-temporal (nt), 2 = write-through scopes (sc0 sc1), 3 = sc0 sc1 nt.
Times the evaluator LAUNCH alone (the library's HIP events around it, gpk_prof_read_assembly) for the four layouts at the BASELINE
sizes, min and median of `reps` runs per variant, variants interleaved; checks that every variant writes the same bits.
Also the chip's plain write ceiling for reference (gpk_ubench_hbm_write, development build).

    python tools/assembly_store_ab.py [--reps 15]          (through gpurun, from the repo root)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonlinpdes-gpsolver_amd'))
import gpk  # noqa: E402

CASES = [  # layout, kernel, parameter, N_domain, N_boundary (effective), BASELINE config
    ('Nonlinear_elliptic', 'Gaussian', 0.2, 4000, 400, 'c2'),
    ('Burgers', 'anisotropic_Gaussian', [0.3, 0.05], 2000, 399, 'c3'),
    ('Darcy_u', 'Gaussian', 0.2, 1600, 200, 'c4 (u)'),
    ('Darcy_a', 'Gaussian', 0.2, 1600, 200, 'c4 (a)'),
    ('Nonlinear_elliptic', 'Gaussian', 0.2, 16000, 2000, 'c5'),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=15)
    a = ap.parse_args()
    ctx = gpk.Context(0, dev=True)
    out = {'device': ctx.device_info(), 'hbm_write_ubench_gbs': ctx.ubench_hbm_write(1 << 30, 10), 'cases': []}
    rng = np.random.RandomState(0)
    for lay, kern, kp, Nd, Nb, name in CASES:
        Xd = rng.uniform(0, 1, (Nd, 2)); Xb = rng.uniform(0, 1, (Nb, 2))
        T = None
        times = {v: [] for v in range(4)}
        ref = None
        same = True
        ctx.prof_enable(True)
        for rep in range(a.reps + 1):
            for v in range(4):
                ctx.tune(55, v)
                T, _ = ctx.assemble(lay, kern, kp, Xd, Xb, 1e-10, 'adaptive', out=T)
                ctx.synchronize()
                ms = ctx.prof_read_assembly()
                if rep > 0:
                    times[v].append(ms)
                elif Nd <= 4000:                                      # first pass: the bits (not at 9 GB)
                    got = T.download()
                    if ref is None:
                        ref = got
                    else:
                        same = same and np.array_equal(ref, got)
        ctx.prof_enable(False)
        ctx.tune(55, 0)
        n = T.rows
        nbytes = 8.0 * n * n
        row = {'case': name, 'layout': lay, 'order': n, 'bytes': nbytes, 'bit_identical': bool(same), 'variants': {}}
        for v, label in enumerate(('plain', 'nt', 'sc0_sc1', 'sc0_sc1_nt')):
            t = np.array(times[v])
            row['variants'][label] = {'min_ms': float(t.min()), 'median_ms': float(np.median(t)),
                                      'gbs_at_min': nbytes / (t.min() * 1e-3) / 1e9, 'gbs_at_median': nbytes / (np.median(t) * 1e-3) / 1e9}
        out['cases'].append(row)
        print(json.dumps(row), flush=True)
        T.free()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'r06_assembly_store_ab.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    ctx.close()


if __name__ == '__main__':
    main()

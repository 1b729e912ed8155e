#!/usr/bin/env python3
"""Per-launch rate of the solve phase (gemm_f64_kernel<NN>) in the LAST Gauss-Newton step of a kernel trace: executed flops from the
launch model of bench.py (mirrored here with the launch list) / duration.  Usage: solve_launches.py <tag> N nz db"""
import csv, glob, re, sys
tag, N, nz, db = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
f = glob.glob(f'/root/repo/gpurun_out/r02c/{tag}/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'gn_build_kernel' in r['Kernel_Name']]
ends = [i for i, r in enumerate(rows) if 'axpy_rev_kernel' in r['Kernel_Name']]
e = ends[-1]; s = [i for i in starts if i < e][-1]
T = []
for r in rows[s:e + 1]:
    m = re.search(r'gemm_f64_kernel<(\d+), (\d+), \d+, \d+, (true|false), (true|false), (true|false)', r['Kernel_Name'])
    if m and m.group(3) == 'false' and m.group(4) == 'false':
        T.append((m.group(1), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
num_cu, nb, bk = 256, 64, 16
nrhs, lead = nz + 1, nz
L = []
def gemm(m, n, k, lz, tri):
    if m <= 0 or n <= 0: return
    t64 = ((m + 63) // 64) * ((n + 63) // 64)
    if tri: t64 //= 2
    bm = 32 if (t64 < 2 * num_cu and m >= 64) else 64
    fl = 0
    for n0 in range(0, n, 64):
        bn = min(64, n - n0)
        k0 = (max(0, lz - (n0 + 64)) // bk) * bk if lz > 0 else 0
        if tri:
            for m0 in range(0, m, bm): fl += 2.0 * min(bm, m - m0) * bn * max(0, min(k, m0 + bm) - k0)
        else:
            fl += 2.0 * m * bn * max(0, k - min(k0, k))
    L.append((m, n, k, lz, tri, fl))
def rec(n, row0):
    if n <= 0: return
    clo = lead - (row0 + n); clo = (clo // nb) * nb if clo > 0 else 0
    if clo >= nrhs: return
    if n <= db:
        gemm(n, nrhs - clo, n, max(lead - row0 - clo, 0), True); return
    n1 = ((n // 2 + db - 1) // db) * db
    if n1 >= n: n1 = db
    rec(n1, row0)
    c1 = lead - (row0 + n1); c1 = (c1 // nb) * nb if c1 > 0 else 0
    if c1 < nrhs: gemm(n - n1, nrhs - c1, n1, max(lead - row0 - c1, 0), False)
    rec(n - n1, row0 + n1)
rec(N, 0)
tot = 0
for l, t in zip(L, T):
    print(l[:5], 'BM', t[0], f'{l[5] / 1e9:9.2f} GF {t[1]:8.1f} us {l[5] / t[1] / 1e6:6.1f} TF/s'); tot += t[1]
print('flops %.1f GF, %.1f us, %d launches (model %d)' % (sum(l[5] for l in L) / 1e9, tot, len(T), len(L)))

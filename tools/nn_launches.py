#!/usr/bin/env python3
"""The solve-phase launches (gemm_f64_kernel<.., NN>) of the LAST Gauss-Newton step of a kernel trace written by tools/trace_c2.sh:
tile configuration, grid, duration.  Usage: nn_launches.py <tag>"""
import csv, glob, re, sys
tag = sys.argv[1]
f = glob.glob(f'/root/repo/gpurun_out/r02c/{tag}/**/*kernel_trace.csv', recursive=True)
f = max(f, key=lambda p: __import__('os').path.getmtime(p))
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'gn_build_kernel' in r['Kernel_Name']]
ends = [i for i, r in enumerate(rows) if 'axpy_rev_kernel' in r['Kernel_Name']]
e = ends[-1]; s = [i for i in starts if i < e][-1]
t0 = int(rows[s]['Start_Timestamp']); tot = 0.0
for r in rows[s:e + 1]:
    m = re.search(r'gemm_f64_kernel<(\d+), (\d+), \d+, \d+, (true|false), (true|false)', r['Kernel_Name'])
    if m and m.group(3) == 'false' and m.group(4) == 'false':
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += d
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  tile {m.group(1)}x{m.group(2)}  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}  {d:8.1f} us")
print(f'sum of NN launches {tot:.1f} us')

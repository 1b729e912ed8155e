#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of bench.py and prints the kernel sequence of the LAST Gauss-Newton step:
per kernel short name, grid, duration, gap to the previous kernel; plus per-name totals.  Usage: trace_step.py <csv> [--all]"""
import csv
import re
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\[clone .*', '', n)
    m = re.match(r'(?:void )?([A-Za-z0-9_]+)(<[^(]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n[:60]


# a step starts at gn_build_kernel
starts = [i for i, r in enumerate(rows) if 'gn_build_kernel' in r['Kernel_Name']]
ends = [i for i, r in enumerate(rows) if 'axpy_rev_kernel' in r['Kernel_Name'] or 'axpy_kernel' in r['Kernel_Name']]
if not starts:
    sys.exit('no gn_build_kernel found')
s = starts[-1] if len(starts) == 1 or starts[-1] < ends[-1] else starts[-2]
e = [i for i in ends if i > s][0]
step = rows[s - 1:e + 1]
t0 = int(step[0]['Start_Timestamp'])
tot = collections.OrderedDict()
prev_end = None
busy = 0
for r in step:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = short(r['Kernel_Name'])
    key = (name, r['Grid_Size_X'] if '--grid' in sys.argv else '')
    d = tot.setdefault(key, [0, 0.0])
    d[0] += 1; d[1] += (b - a) / 1e3
    busy += b - a
    if '--all' in sys.argv:
        print(f"{(a - t0) / 1e3:9.1f} us  +{((a - prev_end) / 1e3 if prev_end else 0):6.1f} gap  {(b - a) / 1e3:8.1f} us  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} x{r['Workgroup_Size_X']:>4}  vgpr {r['VGPR_Count']:>3} lds {r['LDS_Block_Size']:>6}  {name}")
    prev_end = b
span = (int(step[-1]['End_Timestamp']) - t0) / 1e3
print(f'step span {span:.1f} us, kernel-busy {busy / 1e3:.1f} us, {len(step)} launches')
for (name, g), (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f'{us:9.1f} us  {c:4d} x  {us / c:8.1f} avg   {name} {g}')

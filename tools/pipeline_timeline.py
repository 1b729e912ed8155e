#!/usr/bin/env python3
"""Timeline of the pipelined product + factorisation phase of the LAST Gauss-Newton step in a rocprofv3 kernel trace:
GEMM-stream launches one by one, chain-stream kernels summarised per 512-column block.  Usage: pipeline_timeline.py <tag>"""
import csv, glob, re, sys
f = glob.glob(f'/root/repo/gpurun_out/r02c/{sys.argv[1]}/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); m = re.match(r'(?:void )?([A-Za-z0-9_]+)(<[^(]*>)?', n); return (m.group(1) + (m.group(2) or '')) if m else n[:50]
starts = [i for i, r in enumerate(rows) if 'gn_build_kernel' in r['Kernel_Name']]
ends = [i for i, r in enumerate(rows) if 'axpy_rev_kernel' in r['Kernel_Name']]
e = ends[-1]; s = [i for i in starts if i < e][-1]
step = rows[s - 1:e + 1]
t0 = int(step[0]['Start_Timestamp'])
first_tn = [i for i, r in enumerate(step) if re.search(r'gemm_f64_kernel<\d+, \d+, \d+, \d+, true, false', r['Kernel_Name'])][0]
for r in step[first_tn - 1:]:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp']); nm = short(r['Kernel_Name'])
    if 'potrf_panel' in nm or 'k64' in nm: continue
    print(f"{(a - t0) / 1e3:8.1f} -> {(b - t0) / 1e3:8.1f} {(b - a) / 1e3:7.1f}  q{r['Queue_Id']} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d} {nm}")
pk = [r for r in step if 'potrf_panel' in r['Kernel_Name'] or 'k64' in r['Kernel_Name']]
i = 0; blk = 0
while i < len(pk):
    grp = pk[i:i + 15]
    a = int(grp[0]['Start_Timestamp']); b = int(grp[-1]['End_Timestamp'])
    pan = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in grp if 'potrf_panel' in r['Kernel_Name']]
    k64 = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in grp if 'k64' in r['Kernel_Name']]
    print(f'chain blk {blk}: {(a - t0) / 1e3:8.1f} -> {(b - t0) / 1e3:8.1f} span {(b - a) / 1e3:7.1f}  panel avg {sum(pan) / len(pan):5.1f} k64 avg {sum(k64) / max(len(k64), 1):5.1f}')
    i += 15; blk += 1

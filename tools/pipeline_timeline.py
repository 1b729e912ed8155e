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
# ---- per-queue summary of the phase (from the first product launch to the end of the last factorisation kernel): busy time = sum of
#      kernel durations on that queue, span = first start to last end; the chain stream carries the panel kernels and their small updates
phase = step[first_tn:]
last_fact = max(i for i, r in enumerate(phase) if 'potrf_panel' in r['Kernel_Name'])
phase = phase[:last_fact + 1]
p0 = int(phase[0]['Start_Timestamp']); p1 = max(int(r['End_Timestamp']) for r in phase)
print(f'\nphase (first product launch -> last panel kernel): {(p1 - p0) / 1e3:.1f} us')
byq = {}
for r in phase:
    byq.setdefault(r['Queue_Id'], []).append(r)
for q, rs in sorted(byq.items()):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e3
    a = min(int(r['Start_Timestamp']) for r in rs); b = max(int(r['End_Timestamp']) for r in rs)
    npan = sum(1 for r in rs if 'potrf_panel' in r['Kernel_Name'])
    tpan = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs if 'potrf_panel' in r['Kernel_Name']) / 1e3
    tn = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs if re.search(r'gemm_f64_kernel<\d+, \d+, \d+, \d+, true, false', r['Kernel_Name'])) / 1e3
    print(f'queue {q}: {len(rs):4d} kernels, busy {busy:7.1f} us of a {(b - a) / 1e3:7.1f} us span ({(a - p0) / 1e3:7.1f} -> {(b - p0) / 1e3:7.1f}); '
          f'products (TN) {tn:7.1f} us, {npan} panel kernels {tpan:7.1f} us')

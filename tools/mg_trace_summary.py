#!/usr/bin/env python3
"""Summary of a `rocprofv3 --hip-runtime-trace --kernel-trace` run of tools/mg_lookahead_probe.py (MODES=1: look-ahead plan only):
host synchronisations against the number of factorisations and panels (the plan has ONE host read per factorisation), kernels per
stream (three streams: trailing updates, panel, -- no communication stream with one rank), and how much of the panel stream's busy
time overlaps with update kernels on the main stream.  Usage: mg_trace_summary.py <dir with the csv files> N panel nfact"""
import csv, glob, os, sys, collections
d, N, panel, nfact = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
api = glob.glob(os.path.join(d, '**', '*hip_api_trace.csv'), recursive=True)
ker = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
calls = collections.Counter()
for f in api:
    for r in csv.DictReader(open(f)):
        calls[r['Function']] += 1
rows = []
for f in ker:
    rows += list(csv.DictReader(open(f)))
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r.get('Stream_Id', r.get('Queue_Id', '?'))].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
print(f'factorisations {nfact}, panels per factorisation {-(-N // panel)} (x 8 panel kernels of 64 columns each)')
print('host synchronisation calls in the whole process:', {k: v for k, v in calls.items() if 'Synchronize' in k})
print('event calls:', {k: v for k, v in calls.items() if 'Event' in k})
for s, ks in sorted(by_stream.items(), key=lambda kv: -len(kv[1])):
    names = collections.Counter(('panel' if 'potrf_panel' in n else 'gemm' if 'gemm' in n else 'assemble' if 'assemble' in n else 'other') for _, _, n in ks)
    busy = sum(b - a for a, b, _ in ks) / 1e6
    print(f'stream/queue {s}: {len(ks)} kernels {dict(names)}, busy {busy:.1f} ms')
pan = sorted((a, b) for ks in by_stream.values() for a, b, n in ks if 'potrf_panel' in n)
gem = sorted((a, b) for ks in by_stream.values() for a, b, n in ks if 'gemm_f64' in n)
ov = 0; j = 0
for a, b in pan:
    while j < len(gem) and gem[j][1] <= a: j += 1
    k = j
    while k < len(gem) and gem[k][0] < b:
        ov += max(0, min(b, gem[k][1]) - max(a, gem[k][0])); k += 1
tot = sum(b - a for a, b in pan)
print(f'panel kernels: {tot / 1e6:.1f} ms busy, {100.0 * ov / max(tot, 1):.0f} % of it while a trailing-update GEMM is running on another stream')

#!/usr/bin/env python3
"""gpurun_out/traffic (tools/traffic_experiment.sh) -> profiles/<round>_traffic_experiment.json: per GPK_DEBUG_SET variant the phase times of
the config-2 step and, per Gauss-Newton step, the HBM-side fetch bytes (FETCH_SIZE KiB x 2, gfx950 correction for 16-byte-per-lane reads),
the L2 hit rate and the MFMA-busy fraction of the solve-phase launches (gemm_f64_kernel<.., NN>) and of the product launches (<.., TN>)."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'traffic')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'

def kind(name):
    m = re.search(r'gemm_f64_kernel<(\d+), (\d+), \d+, \d+, (true|false), (true|false)', name)
    return None if not m else ('T' if m.group(3) == 'true' else 'N') + ('T' if m.group(4) == 'true' else 'N')

def table(d, name):
    g = glob.glob(os.path.join(d, f'pmc_{name}', '**', '*counter_collection.csv'), recursive=True)
    out = {}
    if not g:
        return out
    for r in csv.DictReader(open(max(g, key=os.path.getmtime))):
        k = (int(r['Dispatch_Id']), r['Kernel_Name'], int(r['Grid_Size']))
        e = out.setdefault(k, {'dur_us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return out

def per_step(t, which, key):
    first = min((k[0] for k in t if 'gn_build_kernel' in k[1]), default=None)
    steps = sum(1 for k in t if 'reverse_copy_kernel' in k[1])
    if first is None or not steps:
        return None
    rows = [v[key] for k, v in t.items() if k[0] > first and kind(k[1]) == which and key in v]
    return sum(rows) / steps if rows else None

res = []
for d in sorted(glob.glob(os.path.join(SRC, 'v*'))):
    v = open(os.path.join(d, 'variant.txt')).read().strip()
    e = {'GPK_DEBUG_SET': v or '(default)'}
    lines = [l for l in open(os.path.join(d, 'bench.json')).read().splitlines() if l.startswith('{')]
    if lines:
        b = json.loads(lines[-1])
        e.update({'steps_per_s': b['value'], 'ms_per_step': b['ms_per_step'], 'phases_ms': b.get('phases_ms')})
    f, t, m = table(d, 'FETCH_SIZE'), table(d, 'TCC_HIT_sum+TCC_MISS_sum'), table(d, 'SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE')
    for which, label in (('NN', 'solve'), ('TN', 'product')):
        fb = per_step(f, which, 'FETCH_SIZE')
        h, ms = per_step(t, which, 'TCC_HIT_sum'), per_step(t, which, 'TCC_MISS_sum')
        bz, ga, du = per_step(m, which, 'SQ_VALU_MFMA_BUSY_CYCLES'), per_step(m, which, 'GRBM_GUI_ACTIVE'), per_step(m, which, 'dur_us')
        e[label] = {'fetch_GB_per_step': None if fb is None else fb * 2048.0 / 1e9,
                    'l2_hit_rate': None if h is None or ms is None else h / (h + ms),
                    'mfma_busy_fraction': None if not bz or not ga else bz / 1024 / (ga / 8),
                    'sum_of_launch_us_per_step_under_pmc': du,
                    'tiles': sorted({re.search(r'gemm_f64_kernel<(\d+, \d+)', k[1]).group(1) for k in f if kind(k[1]) == which})}
    res.append(e)
out = {'what': 'config-2 Gauss-Newton step under GPK_DEBUG_SET tile-configuration variants (tools/traffic_experiment.sh): does less HBM-side traffic buy time?',
       'keys': {'0=1': '128x128 / 16-wave tile for every launch', '33=N': '128x64 / 8-wave tile from N 64x64-tiles on (default 1500)',
                '38=N': '128x128 / 16-wave tile from N 64x64-tiles on (default 6000)', '6=1': 'supertile order of the one-launch product', '12=0': 'one-launch product, no pipeline'},
       'variants': res}
p = os.path.join(ROOT, 'profiles', f'{tag}_traffic_experiment.json')
open(p, 'w').write(json.dumps(out, indent=1) + '\n')
print(json.dumps(out, indent=1))

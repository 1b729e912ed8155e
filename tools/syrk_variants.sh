#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "12=0" "12=0,42=2" "12=0,0=4" "12=0,0=4,42=2" "12=0,0=4,42=2,46=0"; do
  export GPK_DEBUG_SET=$v
  python bench.py --no-cpu-baseline --no-n10k --no-sharded-config --no-structured --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']; r=d['roofline_syrk']; print('$v', round(d['ms_per_step'],3), 'solve', round(p['trsm'],3), 'phase', round(p['syrk_and_potrf_H'],3), 'syrk launch', round(p['syrk_launches_sum'],3), 'frac', round(r['frac'],3), d['l2_error']['pts_L2_err'])"
done

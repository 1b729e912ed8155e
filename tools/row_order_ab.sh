#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "49=0" "49=1" "49=0" "49=1" "49=1,24=0" "49=0,24=0"; do
  export GPK_DEBUG_SET=$v
  python bench.py --no-cpu-baseline --no-n10k --no-sharded-config --no-structured --steps 12 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']; print('$v', round(d['ms_per_step'],3), 'solve', round(p['trsm'],3), 'phase', round(p['syrk_and_potrf_H'],3), 'syrk_sum', round(p['syrk_launches_sum'],3), d['l2_error']['pts_L2_err'])"
done

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {
  python bench.py --no-cpu-baseline --no-n10k --no-sharded-config --no-structured --steps 12 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']; print('$1', round(d['ms_per_step'],3), 'solve', round(p['trsm'],3), 'phase', round(p['syrk_and_potrf_H'],3), 'trsv', round(p['trsv_update'],3))"
}
unset GPK_DEBUG_SET; run none
GPK_DINV_BLOCK=1024 run dinv1024
export GPK_DEBUG_SET=33=1000; run 33=1000
export GPK_DEBUG_SET=33=2500; run 33=2500
export GPK_DEBUG_SET=33=0; run 33=0
export GPK_DEBUG_SET=8=0; run 8=0
unset GPK_DEBUG_SET; run none

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in none "38=3000" "38=1500" "38=12000" "35=128" "35=256" "33=800" ; do
  if [ "$v" = "none" ]; then unset GPK_DEBUG_SET; else export GPK_DEBUG_SET=$v; fi
  python bench.py --workload n10k --no-cpu-baseline --no-sharded-config --no-structured --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']; print('$v', round(d['ms_per_step'],2), 'solve', round(p['trsm'],2), 'phase', round(p['syrk_and_potrf_H'],2), 'syrk', round(p['syrk_launches_sum'],2), 'chol_theta', round(d['one_time_ms']['cholesky_theta'],1))"
done

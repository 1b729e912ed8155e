#!/bin/bash
# A/B of a bench workload under GPK_DEBUG_SET variants (compact line: phases_ms, one_time_ms):
#   tools/ab_bench.sh c2 "" "24=1000,26=1" ...      workloads: c2 (default), c3, c4, n10k
wl=$1; shift
case $wl in
  c2) FLAGS="--no-sharded-config --no-cpu-baseline --no-structured --no-n10k --no-c3c4";;
  n10k) FLAGS="--workload n10k --no-sharded-config --no-cpu-baseline --no-structured";;
  *) FLAGS="--workload $wl --no-cpu-baseline";;
esac
for v in "$@"; do
  for rep in 1 2; do
  GPK_DEBUG_SET="$v" GPK_BENCH_DETAIL_DIR=/tmp timeout 300 python3 bench.py --steps 8 --warmup 3 $FLAGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl [$v]', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), d.get('phases_ms'), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (d.get('one_time_ms') or {}).items()}, flush=True)"
  done
done

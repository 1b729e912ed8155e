# A/B of the config-2 bench under GPK_DEBUG_SET variants: ab_bench.sh "24=1000,26=0" "24=1000,26=1" ...
for v in "$@"; do
  for rep in 1 2; do
  GPK_DEBUG_SET="$v" timeout 120 python3 bench.py --steps 8 --warmup 3 --no-sharded-config --no-cpu-baseline --no-structured 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3), d.get('phases_ms_per_step'), d.get('pts_L2_err'))"
  done
done

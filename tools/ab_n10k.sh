# A/B of the north-star size (N_d = 10000) under GPK_DEBUG_SET variants
for v in "$@"; do
  GPK_DEBUG_SET="$v" timeout 300 python3 bench.py --workload n10k --steps 4 --warmup 2 --no-sharded-config --no-cpu-baseline --no-structured 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],2), round(d['ms_per_step'],2), d.get('phases_ms_per_step'))"
done

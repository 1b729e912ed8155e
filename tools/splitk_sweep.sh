# config-2 bench under different split-K targets of the pipelined products (gpk_debug_set key 24): prints value + phases
for u in 0 1000 1500 2000 3000; do
  echo "units=$u"
  GPK_DEBUG_SET="24=$u" timeout 120 python3 bench.py --steps 8 --warmup 3 --no-sharded-config --no-cpu-baseline --no-structured 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('phases_ms', d.get('phases')))"
done

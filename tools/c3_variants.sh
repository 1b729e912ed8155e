#!/bin/bash
# BASELINE config 3 (Burgers, Hessian of order 6001: the largest size that still takes the two-partition pipeline) under the pipeline's
# switches: per-step time and phases.  Run through gpurun from the repo root.
for v in "" "12=0" "24=2000" "24=500" "34=128" "34=0" "13=64" "14=5000" "28=256,29=256" "26=1"; do
  echo "== GPK_DEBUG_SET=$v"
  GPK_DEBUG_SET=$v python bench.py --workload c3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,3) if isinstance(v,float) else v for k,v in d['phases_ms_per_step'].items()}, 'syrk frac', round(d['roofline_syrk']['frac'],3))"
done

#!/bin/bash
# Two ranks of bench.py on ONE GPU (process group over gloo instead of RCCL, which refuses two ranks on one device): exercises
# the N > 1 control flow of the benchmark with the real kernels -- replicas of config 2, then the sharded config 5 with BOTH
# executors of the plan: the native one (gpk_mg_*, collectives = host-staged stand-ins bound to the ncclBroadcast /
# ncclAllGather entry points) and the Python one over torch.distributed.  Run through gpurun from the repo root.
PORT=29533
for engine in native python; do
  for r in 0 1; do
    RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT GPK_BENCH_BACKEND=gloo GPK_BENCH_COMM=staged GPK_BENCH_SHARDED=$engine \
      timeout 1200 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/bench2_${engine}_rank$r.log 2>&1 &
  done
  wait
  echo "== executor: $engine"
  grep "^{" gpurun_out/bench2_${engine}_rank0.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); sc=d.get('sharded_config'); print('value', d['value'], 'n_gpus', d['n_gpus'], d['scaling'], d['ms_per_step'], d['l2_error']['pts_L2_err']); print('sharded', {k: sc.get(k) for k in ('value','n_gpus','ms_per_step','f1_tflops','one_time_ms','error','mode_probe')}, sc.get('l2_error'), sc.get('config', {}).get('executor'))"
  tail -2 gpurun_out/bench2_${engine}_rank1.log
  PORT=$((PORT+1))
done

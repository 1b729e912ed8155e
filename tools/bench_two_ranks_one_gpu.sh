#!/bin/bash
# Two ranks of bench.py on ONE GPU (process group over gloo instead of RCCL, which refuses two ranks on one device): exercises
# the N > 1 control flow of the benchmark with the real kernels -- replicas of config 2 (secondary), then the sharded config 5 (= `value`
# since round 4, with the same configuration on rank 0 alone beside it) with BOTH
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
  grep "^{" gpurun_out/bench2_${engine}_rank0.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('value (sharded config 5)', d['value'], 'n_gpus', d['n_gpus'], d['scaling'], 'ms/step', d['ms_per_step'], 'pts_L2_err', d['l2_error']['pts_L2_err'])
print('vs_1gpu', d.get('vs_1gpu'), 'one GPU in the same job:', (d.get('one_gpu_same_job') or {}).get('ms_per_step'), 'executor:', d['config'].get('executor'))
print('line bytes', len(json.dumps(d, separators=(',', ':'))), 'value_workload', d.get('value_workload'), 'preflight', d.get('preflight'))
print('mode_probe', d.get('mode_probe')); print('parity', {k: (d.get('parity') or {}).get(k) for k in ('z1_rel_dev_vs_B2', 'ok', 'skipped')}, 'parity_failed', d.get('parity_failed'))
print('replicas_c2', {k: (d.get('replicas_c2') or {}).get(k) for k in ('value', 'n_gpus', 'scaling', 'ms_per_step', 'error')})"
  tail -2 gpurun_out/bench2_${engine}_rank1.log
  PORT=$((PORT+1))
done

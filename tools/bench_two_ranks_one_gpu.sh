#!/bin/bash
# two ranks of bench.py on ONE GPU (gloo instead of RCCL): exercises the N > 1 control flow with the real kernels
PORT=29533
for r in 0 1; do
  RANK=$r WORLD_SIZE=2 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT GPK_BENCH_BACKEND=gloo \
    timeout 800 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/bench2_rank$r.log 2>&1 &
done
wait
grep "^{" gpurun_out/bench2_rank0.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); sc=d.get('sharded_config'); print('value', d['value'], 'n_gpus', d['n_gpus'], d['scaling'], d['ms_per_step'], d['l2_error']['pts_L2_err']); print('sharded', {k: sc.get(k) for k in ('value','n_gpus','ms_per_step','f1_tflops','one_time_ms','error')}, sc.get('l2_error'))"
tail -3 gpurun_out/bench2_rank1.log

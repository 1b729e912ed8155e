cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
for f in test_gpu_mg test_gpu_sharded_ops test_gpu_sharded_world2; do
  timeout 900 python -m pytest tests/$f.py -x -q -m gpu > gpurun_out/r06g/$f.txt 2>&1; echo "$f rc $?" >> gpurun_out/r06g/summary.txt
done
timeout 1500 python -m pytest tests/test_gpu_mg.py tests/test_gpu_sharded_ops.py tests/test_gpu_sharded_world2.py -x -q -m gpu > gpurun_out/r06g/together.txt 2>&1; echo "together rc $?" >> gpurun_out/r06g/summary.txt
timeout 1500 python -m pytest tests/test_gpu_sharded_ops.py tests/test_gpu_sharded_world2.py -x -q -m gpu > gpurun_out/r06g/two.txt 2>&1; echo "ops+world2 rc $?" >> gpurun_out/r06g/summary.txt
cat gpurun_out/r06g/summary.txt

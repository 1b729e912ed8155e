cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06i
timeout 1200 python -m pytest tests/test_gpu_structured.py tests/test_gpu_mg.py tests/test_gpu_sharded_ops.py -x -q -m gpu > gpurun_out/r06i/tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06i/tests.txt
timeout 600 python bench.py --workload c3 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06i/c3.json 2> gpurun_out/r06i/c3.err
timeout 600 python bench.py --workload c4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06i/c4.json 2> gpurun_out/r06i/c4.err
tail -12 gpurun_out/r06i/tests.txt | cut -c1-300

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
timeout 1500 python -m pytest tests/test_gpu_mg.py tests/test_gpu_sharded_ops.py tests/test_gpu_sharded_world2.py -x -q -m gpu > gpurun_out/r06f/tests_mg.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06f/tests_mg.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize_oracle.py -x -q -m gpu -k "config3 or config4" -s > gpurun_out/r06f/tests_full.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06f/tests_full.txt
GPK_BENCH_SECONDARY_CPU=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06f/bench.json 2> gpurun_out/r06f/bench.err
tail -5 gpurun_out/r06f/tests_mg.txt; grep "structured step\|passed\|failed" gpurun_out/r06f/tests_full.txt | tail -5

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 1200 python -m pytest tests/test_gpu_bench_line.py::test_bare_two_rank_command_on_one_gpu tests/test_gpu_structured.py tests/test_gpu_solver_api.py tests/test_gpu_mg.py -x -q -m gpu > gpurun_out/r06a/tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06a/tests.txt
GPK_BENCH_SECONDARY_CPU=0 timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; echo "bench rc $?" >> gpurun_out/r06a/tests.txt
cp bench_detail.json gpurun_out/r06a/bench_detail.json
rocprofv3 --kernel-trace --stats -d gpurun_out/r06a/c4prof -o c4 -- python3 bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06a/c4.json 2> gpurun_out/r06a/c4.err
tail -3 gpurun_out/r06a/tests.txt

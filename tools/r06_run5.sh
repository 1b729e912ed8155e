cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06e/tests_all.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06e/tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06e/smoke.txt 2>&1; echo "smoke rc $?" >> gpurun_out/r06e/smoke.txt
GPK_BENCH_BACKEND=gloo GPK_BENCH_COMM=staged GPK_SHARDED_TIMEOUT=1200 timeout 1500 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r06e/bench_2ranks_one_gpu.json 2> gpurun_out/r06e/bench_2ranks_one_gpu.err; echo "bench2 rc $?" >> gpurun_out/r06e/smoke.txt
tail -4 gpurun_out/r06e/tests_all.txt; cat gpurun_out/r06e/smoke.txt | tail -4

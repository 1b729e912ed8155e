cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests/test_gpu_mg.py tests/test_gpu_variants.py tests/test_gpu_structured.py tests/test_gpu_sharded_world2.py -x -q -m gpu > gpurun_out/r06d/tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06d/tests.txt
timeout 1500 python -m pytest tests/test_gpu_bench_line.py -x -q -m gpu -k "bare" > gpurun_out/r06d/tests_bare.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06d/tests_bare.txt
timeout 600 python bench.py --workload c4 --steps 8 --warmup 2 --no-cpu-baseline --no-structured > gpurun_out/r06d/c4.json 2> gpurun_out/r06d/c4.err
tail -6 gpurun_out/r06d/tests.txt; tail -12 gpurun_out/r06d/tests_bare.txt

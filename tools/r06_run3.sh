cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
timeout 1200 python -m pytest tests/test_gpu_structured.py -x -q -m gpu > gpurun_out/r06c/tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06c/tests.txt
timeout 600 python bench.py --workload c3 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06c/c3.json 2> gpurun_out/r06c/c3.err
timeout 600 python bench.py --workload c4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06c/c4.json 2> gpurun_out/r06c/c4.err
tail -15 gpurun_out/r06c/tests.txt

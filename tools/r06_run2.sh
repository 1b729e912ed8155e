cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "darcy" > gpurun_out/r06b/tests.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06b/tests.txt
timeout 600 python tools/assembly_store_ab.py --reps 15 > gpurun_out/r06b/asm_ab.txt 2>&1
timeout 600 python bench.py --workload c4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06b/c4.json 2> gpurun_out/r06b/c4.err
GPK_DARCY_CACHE=0 timeout 600 python bench.py --workload c4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r06b/c4_nocache.json 2> gpurun_out/r06b/c4_nocache.err
tail -3 gpurun_out/r06b/tests.txt; tail -5 gpurun_out/r06b/asm_ab.txt | cut -c1-600

cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-sharded-config --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02c/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02c/err.txt
ls -R $GRAFT_REPO_ROOT/gpurun_out/r02c | head

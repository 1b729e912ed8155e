cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r02d
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02d -- python3 $GRAFT_REPO_ROOT/tools/overlap_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r02d/out.txt 2>&1

# kernel trace of the north-star size (development aid): gpurun_out/r02c/<tag>/
cd /tmp && export TMPDIR=/tmp
TAG=${1:-n10k}
export GPK_DEBUG_SET=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --workload n10k --steps 2 --warmup 1 --no-sharded-config --no-cpu-baseline --no-structured > $OUT/bench.json 2> $OUT/err.txt

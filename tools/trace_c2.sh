# kernel trace of the config-2 bench (development aid): gpurun_out/r02c/<tag>/ ; usage: trace_c2.sh <tag> [GPK_DEBUG_SET value]
cd /tmp && export TMPDIR=/tmp
TAG=${1:-default}
export GPK_DEBUG_SET=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02c/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-sharded-config --no-cpu-baseline --no-structured --no-n10k --no-c3c4 > $OUT/bench.json 2> $OUT/err.txt

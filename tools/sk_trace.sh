#!/bin/bash
# per-launch kernel traces of one bench run per variant (GPK_DEBUG_SET pairs given as arguments, "none" = defaults) -> gpurun_out/sk_trace/<tag>.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sk_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    tag=$(echo $v | tr '=,' '__')
    if [ "$v" = "none" ]; then unset GPK_DEBUG_SET; else export GPK_DEBUG_SET=$v; fi
    rm -rf /tmp/skt_$tag
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/skt_$tag -- python3 $REPO/bench.py --no-cpu-baseline --no-n10k --no-sharded-config --no-structured --steps 4 --warmup 2 > $OUT/$tag.json 2> $OUT/$tag.err
    f=$(find /tmp/skt_$tag -name "*kernel_trace.csv" | head -1)
    python3 $REPO/tools/trace_step.py $f --all > $OUT/$tag.txt 2>&1
done

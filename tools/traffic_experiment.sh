#!/bin/bash
# Round 5, verdict item 5: does a larger tile / another tile order cut the 12x (solve) / 22x (product) HBM-side traffic of the config-2 step,
# and does it make the step faster?  Per GPK_DEBUG_SET variant: the bench line (phases, no profiler), a FETCH_SIZE pass and a TCC hit/miss
# pass (each its own run, --kernel-trace only).  tools/summarize_traffic.py -> profiles/rNN_traffic_experiment.json
#   tools/traffic_experiment.sh "" "0=1" "33=500" "33=500,38=1000"
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/traffic
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp GPK_BENCH_DETAIL_DIR=/tmp
ONE="--no-sharded-config --no-cpu-baseline --no-structured --no-n10k --no-c3c4"
i=0
for v in "$@"; do
    d=$OUT/v$i; mkdir -p $d; echo "$v" > $d/variant.txt
    export GPK_DEBUG_SET="$v"
    timeout 300 python3 $REPO/bench.py $ONE --steps 10 --warmup 3 > $d/bench.json 2> $d/bench.err
    for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        name=$(echo $pass | tr ' ' '+')
        timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d/pmc_$name -- python3 $REPO/bench.py $ONE --steps 3 --warmup 1 > $d/pmc_$name.json 2> $d/pmc_$name.err
    done
    i=$((i+1))
done
unset GPK_DEBUG_SET
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT

#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. bench.py (default flags) plain -> gpurun_out/prof/bench.json
#   2. the same command under rocprofv3 --kernel-trace --stats -> kernel_stats.csv
#   3. PMC passes, each in its own run with --kernel-trace only (MI355X_MICROARCH.md HBM/rocprofv3 section):
#        a) FETCH_SIZE   b) WRITE_SIZE   c) SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   d) TCC_HIT_sum TCC_MISS_sum
#      (config 2 only: --no-sharded-config, so that the per-kernel averages are those of the value's workload)
# tools/summarize_profiles.py turns gpurun_out/prof into profiles/rNN_*.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $REPO/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
# the value's workload alone (config 2): per-kernel averages here are directly comparable with bench.py's roofline object
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $REPO/bench.py --no-sharded-config --no-cpu-baseline --no-structured > $OUT/bench_c2_under_rocprof.json 2> $OUT/stats_c2.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $pass | tr ' ' '+')
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py --no-cpu-baseline --no-sharded-config --no-structured --steps 3 --warmup 1 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
# keep the merge-back small: only the stats and counter tables
find $OUT -name "*kernel_trace.csv" -path "*stats*" -delete
ls -R $OUT | head -40

#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. bench.py (default flags) plain -> gpurun_out/prof/bench.json
#   2. the same command under rocprofv3 --kernel-trace --stats -> kernel_stats.csv
#   3. the value's workload alone (config 2) under --kernel-trace --stats; the same for configs 3, 4 and the north-star size
#   4. PMC passes at config 2 AND at the north-star size, each in its own run with --kernel-trace only (MI355X_MICROARCH.md
#      HBM/rocprofv3 section):  a) FETCH_SIZE   b) WRITE_SIZE   c) SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE   d) TCC_HIT_sum TCC_MISS_sum
#   5. the Cholesky factorisation at the north-star size (N_domain = 10000, order 21000): the MFMA-busy pass of (4c) at that size
#      -> what the trailing-update launches (gemm_f64_kernel<.., NT>) and the panel kernels do
# tools/summarize_profiles.py / tools/summarize_potrf_pmc.py turn gpurun_out/prof into profiles/rNN_*.
# Every profiled command is `python3 bench.py ...` itself (no env / shell hop between rocprofv3 and the program).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ONE="--no-sharded-config --no-cpu-baseline --no-structured --no-n10k --no-c3c4"
# (the full result object of every bench run goes to GPK_BENCH_DETAIL_DIR: the plain run's is kept, the profiled runs' go to /tmp)
GPK_BENCH_DETAIL_DIR=$OUT timeout 1200 python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
export GPK_BENCH_DETAIL_DIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
# the value's workload alone (config 2): per-kernel averages here are directly comparable with bench.py's roofline object
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $REPO/bench.py $ONE > $OUT/bench_c2_under_rocprof.json 2> $OUT/stats_c2.err
for wl in c3 c4; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python3 $REPO/bench.py --workload $wl --no-cpu-baseline --no-structured > $OUT/bench_${wl}_under_rocprof.json 2> $OUT/stats_$wl.err
done
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $pass | tr ' ' '+')
    timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py $ONE --steps 3 --warmup 1 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
done
# north-star size: factorisation of Theta (order 21000) + Gauss-Newton steps; plain kernel trace with stats, then the four PMC passes
N10K="--workload n10k --no-sharded-config --no-cpu-baseline --no-structured"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_n10k -- python3 $REPO/bench.py $N10K --steps 2 --warmup 1 > $OUT/bench_n10k_under_rocprof.json 2> $OUT/stats_n10k.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $pass | tr ' ' '+')
    timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_n10k_$name -- python3 $REPO/bench.py $N10K --steps 2 --warmup 1 > $OUT/pmc_n10k_$name.json 2> $OUT/pmc_n10k_$name.err
done
# BASELINE configs 3, 4 and 5-on-one-GPU: the same four PMC passes (round 5: no roofline.traffic of the line stays null)
for wl in c3 c4 c5; do
    EXTRA="--no-cpu-baseline --no-structured"; [ $wl = c5 ] && EXTRA="--no-cpu-baseline --no-sharded-parity"
    [ $wl = c5 ] && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python3 $REPO/bench.py --workload $wl $EXTRA --steps 2 --warmup 1 > $OUT/bench_${wl}_under_rocprof.json 2> $OUT/stats_$wl.err
    for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
        name=$(echo $pass | tr ' ' '+')
        timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_${wl}_$name -- python3 $REPO/bench.py --workload $wl $EXTRA --steps 2 --warmup 1 > $OUT/pmc_${wl}_$name.json 2> $OUT/pmc_${wl}_$name.err
    done
done
# (tools/summarize_potrf_pmc.py reads the MFMA-busy pass of the factorisation from pmc_n10k_SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE)
# keep the merge-back small: only the stats and counter tables
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT | head -60

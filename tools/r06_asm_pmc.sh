# PMC passes of the Gram evaluator alone (tools/assembly_store_ab.py, 3 repetitions per store policy): write requests and write-request stalls
# of the L2 <-> fabric interface next to the bytes written (each counter set in its own run, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/asm_pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA[0-9A-Z_]*WRREQ[A-Za-z0-9_]*\|TCC_EA[0-9A-Z_]*WR_[A-Za-z0-9_]*" | sort -u > $OUT/avail.txt
for pass in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $REPO/tools/assembly_store_ab.py --reps 3 > $OUT/$name.log 2>&1
done
find $OUT -name "*agent_info.csv" -delete
ls $OUT; head -30 $OUT/avail.txt

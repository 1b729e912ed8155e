# final verification of the round: the driver's three commands (GPU test-suite, smoke, default bench) and the profile collection
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06z
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06z/tests_all.txt 2>&1; echo "tests rc $?" >> gpurun_out/r06z/tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06z/smoke.txt 2>&1; echo "smoke rc $?" >> gpurun_out/r06z/smoke.txt
bash tools/profile_round.sh > gpurun_out/r06_profile_call.log 2>&1
tail -3 gpurun_out/r06z/tests_all.txt; tail -2 gpurun_out/r06z/smoke.txt

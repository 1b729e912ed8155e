bash tools/ab_bench.sh "27=1" "27=2"
export GPU_MAX_HW_QUEUES=8
echo "GPU_MAX_HW_QUEUES=8"
bash tools/ab_bench.sh "27=0" "27=1" "27=2"

# which kernels of the bench run on which hardware queue (rocprofv3 kernel trace, tools/queue_map.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/qmap
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qmap -o good -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/qmap/good.log 2>&1
python tools/queue_map.py gpurun_out/qmap/good_kernel_trace.csv > gpurun_out/qmap/good_map.txt 2>&1
rm -f gpurun_out/qmap/good_kernel_trace.csv
cat gpurun_out/qmap/good_map.txt

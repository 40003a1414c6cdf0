cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3q_bench -o b -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3q_bench.log 2>&1 || echo fail
python tools/iter_timeline.py gpurun_out/r3q_bench/b_kernel_trace.csv > gpurun_out/r3q_bench_timeline.txt; cat gpurun_out/r3q_bench_timeline.txt | cut -c1-220
python tools/frame_gap_timeline.py gpurun_out/r3q_bench/b_kernel_trace.csv > gpurun_out/r3q_bench_gaps.txt; cat gpurun_out/r3q_bench_gaps.txt
rm -f gpurun_out/r3q_bench/b_kernel_trace.csv
